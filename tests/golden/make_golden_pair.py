#!/usr/bin/env python3
"""PROVISIONAL goldens for the inter-SUNK distance stage: runs the reference's own
workflow/scripts/process-by-contig_lowmem_AR.py (unmodified, where it lies under /root/reference) in the build
container, under pandas 2.x and with the graph_tool / seaborn stand-ins of tests/golden/_shim/ on PYTHONPATH
(neither package exists in the image; the pinned environment is pandas 1.3.4 / graph-tool 2.43,
workflow/envs/viz.yaml:82-100).  Everything up to the graph calls (pandas filtering, pdist, masks, the
multi-position clean-up) is the reference's code running for real; vertex order and the largest-component
tie rule are the stand-in's.  Inputs = per-contig slices of the diag-filtered .sunkpos goldens."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SCRIPT = "/root/reference/workflow/scripts/process-by-contig_lowmem_AR.py"


def run_case(src_case, out_name, bump_rlen, bad_every):
    src = os.path.join(HERE, src_case)
    out = os.path.join(HERE, out_name)
    os.makedirs(out, exist_ok=True)
    rows = [ln for ln in open(os.path.join(src, "expected.diag2")).read().splitlines() if ln]
    by_contig = {}
    for ln in rows:
        by_contig.setdefault(ln.split("\t")[2], []).append(ln)
    rl = open(os.path.join(src, "expected.rlen")).read().splitlines()
    if bump_rlen:
        rl = [f"{x.split(chr(9))[0]}\t{int(x.split(chr(9))[1]) + 10000}" for x in rl]
    open(os.path.join(out, "rlen"), "w").write("\n".join(rl) + "\n")
    meta = {"source": src_case, "contigs": [], "note": "provisional: reference script under pandas 2.x + graph_tool stand-in"}
    for i, (contig, lines) in enumerate(sorted(by_contig.items())):
        ids = sorted({int(x.split("\t")[4]) for x in lines})
        bad = "".join(f"{contig}:{v}\n" for v in ids[::bad_every]) if bad_every else ""
        tag = f"c{i}"
        open(os.path.join(out, f"{tag}.sunkpos"), "w").write("\n".join(lines) + "\n")
        open(os.path.join(out, f"{tag}.bad"), "w").write(bad)
        open(os.path.join(out, f"{tag}.loc"), "w").write(f"{contig}\t0\tACGT\t0\n")
        tsv, bed = os.path.join(out, f"{tag}.expected.tsv"), os.path.join(out, f"{tag}.expected.bed")
        for f in (tsv, bed):
            if os.path.exists(f):
                os.remove(f)
        env = dict(os.environ, PYTHONPATH=os.path.join(HERE, "_shim"), MPLBACKEND="Agg")
        p = subprocess.run([sys.executable, REF_SCRIPT, os.path.join(out, f"{tag}.loc"), os.path.join(out, f"{tag}.sunkpos"),
                            os.path.join(out, "rlen"), os.path.join(out, f"{tag}.bad"), tsv, bed],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        meta["contigs"].append({"tag": tag, "contig": contig, "rc": p.returncode, "rows": len(lines),
                                "has_bed": os.path.exists(bed), "stderr_tail": p.stderr.decode()[-300:] if p.returncode else ""})
    json.dump(meta, open(os.path.join(out, "case.json"), "w"), indent=1, sort_keys=True)
    return meta


def synth_case(out_name, seed, n_reads):
    """Synthetic reads with chimeric jumps, the same SUNK seen at several read positions, equal starts
    (ratio = inf) and reads shorter than 10 kb: exercises the clean-up and component logic."""
    import numpy as np

    out = os.path.join(HERE, out_name)
    os.makedirs(out, exist_ok=True)
    rng = np.random.default_rng(seed)
    rows, rlen = [], []
    for r in range(n_reads):
        n = int(rng.integers(1, 50))
        L = int(rng.choice([5000, 20000, 80000]))
        rlen.append(f"read{r:03d}\t{L}")
        base = int(rng.integers(0, 10**6))
        orient = 1 if rng.random() < 0.5 else -1
        p, seen = 0, []
        for i in range(n):
            p += int(rng.integers(1, 2000))
            true_start = base + orient * p + (3 * 10**6 if orient < 0 else 0)
            u = rng.random()
            if u < 0.1:
                s = int(rng.integers(0, 4 * 10**6))
            elif u < 0.25 and seen:
                s = seen[int(rng.integers(0, len(seen)))]
            else:
                s = true_start + int(rng.integers(-40, 40))
            seen.append(s)
            rows.append((f"read{r:03d}", p, "ctgX", s, (s // 3000) * 3000))
    rows.sort(key=lambda t: (int(t[0][4:]) % 5, t[0], t[1]))
    open(os.path.join(out, "expected.diag2"), "w").write("".join("\t".join(map(str, t)) + "\n" for t in rows))
    open(os.path.join(out, "expected.rlen"), "w").write("\n".join(rlen) + "\n")


def synth_long_case(out_name, seed):
    """Reads with 10^2 .. 10^3 hits (long reads on a diploid assembly), TIE-FREE by construction
    (tests/golden/tie_census.py proves it on the result): every hit is its own SUNK group (ID = start) except planted
    re-sightings of an ID at a far-away read position, which are consistent with almost no other hit and so lose the
    `value_counts` vote by a wide margin; spurious locations stay isolated, so one component dominates."""
    import numpy as np

    out = os.path.join(HERE, out_name)
    os.makedirs(out, exist_ok=True)
    rng = np.random.default_rng(seed)
    rows, rlen = [], []
    for r, n in enumerate((100, 150, 220, 300, 420, 560, 700, 850, 1000, 130, 90, 12)):
        rlen.append(f"long{r:02d}\t{500000 if n > 20 else 9000}")
        base = int(rng.integers(10**6, 10**7))
        orient = 1 if r % 2 == 0 else -1
        p = np.sort(rng.choice(np.arange(1, 450000), size=n, replace=False))
        for i, pi in enumerate(p.tolist()):
            u = rng.random()
            if u < 0.06:
                s = int(rng.integers(0, 3 * 10**7))                       # spurious location
            else:
                s = base + orient * pi + int(rng.integers(-25, 25)) + (2 * 10**7 if orient < 0 else 0)
            rows.append((f"long{r:02d}", pi, "ctgL", s, s))
            if u > 0.95 and i > 10:                                       # the same SUNK "seen" again, somewhere wrong
                rows.append((f"long{r:02d}", int(p[i - 10]) + 7, "ctgL", s + 1, s))
    rows.sort(key=lambda t: (t[0], t[1]))
    open(os.path.join(out, "expected.diag2"), "w").write("".join("\t".join(map(str, t)) + "\n" for t in rows))
    open(os.path.join(out, "expected.rlen"), "w").write("\n".join(rlen) + "\n")


ALL_CASES = (("amy", "pair_amy", False, 7), ("multi_k20", "pair_multi_k20", True, 0), ("multi_k25", "pair_multi_k25", True, 5),
             ("pair_synth_src", "pair_synth", False, 11), ("pair_long_src", "pair_long", False, 37))


if __name__ == "__main__":
    if not os.path.exists(REF_SCRIPT):
        sys.exit("reference tree not present")
    if "--check" in sys.argv:  # regenerate into a temp dir and compare with the committed files
        import filecmp
        import shutil
        import tempfile

        keep = HERE
        with tempfile.TemporaryDirectory() as t:
            for d in ("amy", "multi_k20", "multi_k25"):
                shutil.copytree(os.path.join(keep, d), os.path.join(t, d))
            shutil.copytree(os.path.join(keep, "_shim"), os.path.join(t, "_shim"))
            HERE = t
            synth_case("pair_synth_src", 9, 150)
            synth_long_case("pair_long_src", 26)
            bad = []
            for args in ALL_CASES:
                run_case(*args)
                for f in os.listdir(os.path.join(t, args[1])):
                    if f != "case.json" and not filecmp.cmp(os.path.join(t, args[1], f), os.path.join(keep, args[1], f), shallow=False):
                        bad.append(f"{args[1]}/{f}")
        print("pair goldens reproduce" if not bad else f"MISMATCH: {bad}")
        sys.exit(1 if bad else 0)
    synth_case("pair_synth_src", 9, 150)
    synth_long_case("pair_long_src", 26)
    for args in ALL_CASES:
        m = run_case(*args)
        print(args[1], [(c["tag"], c["rc"], c["has_bed"]) for c in m["contigs"]])
