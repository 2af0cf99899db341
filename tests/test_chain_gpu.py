"""The device-resident second half of the engine (csrc/gs_chain.hip): hit list in HBM -> rows -> diagonal filter -> bad SUNKs ->
pair stage -> union, against the file-to-file path (the drop-in command lines, themselves checked against the oracle and the
reference's own outputs in test_pipeline_gpu.py / test_diag_gpu.py / test_pair_golden.py) and against the oracle directly."""
import io

import numpy as np
import pytest

from gavisunk_amd import (_lib, badsunks, define_sunks, diag_filter_step2, diag_filter_v3, engine, kmerpos_annot3,
                          process_by_contig, synth)
from oracle import oracle

pytestmark = pytest.mark.gpu


def _assembly(rng, n_contigs, length):
    """Two haplotypes of n_contigs contigs each: shared ancestry (so that reads of one hit SUNKs of the other), pasted repeats."""
    fam = rng.integers(0, 4, 1200, dtype=np.uint8)
    haps = [[], []]
    for c in range(n_contigs):
        base = rng.integers(0, 4, length, dtype=np.uint8)
        for h in range(2):
            s = base.copy()
            snp = rng.random(len(s)) < 0.003
            s[snp] = rng.integers(0, 4, int(snp.sum()), dtype=np.uint8)
            for _ in range(4):
                p = int(rng.integers(0, len(s) - 1200))
                s[p:p + 1200] = fam
            haps[h].append((f"h{h + 1}c{c}", s))
    return haps


def test_device_chain_equals_file_chain(tmp_path):
    rng = np.random.default_rng(2024)
    d = tmp_path
    haps = _assembly(rng, 3, 60_000)
    for h in (0, 1):
        (d / f"hap{h + 1}.fa").write_bytes(b"".join(b">%s\n%s\n" % (n.encode(), synth.codes_to_ascii(s)) for n, s in haps[h]))
        (d / f"hap{h + 1}.fa.fai").write_text("".join(f"{n}\t{len(s)}\t1\t2\t3\n" for n, s in haps[h]))
    assert define_sunks.main(["-k", "20", str(d / "hap1.fa"), str(d / "hap2.fa"), "--db", str(d / "db"), "--loc", str(d / "loc")]) == 0
    db = engine.SunkDB(str(d / "db"), str(d / "loc"))
    table = engine.DeviceTable.from_db(db)
    lr = engine.LocRows.from_db(db)
    key_contig = lr.key_contig()
    sets, names_of, lens_of, step2_text = [], [], [], []
    for h in (0, 1):
        seqs = [s for _, s in haps[h]]
        lens = synth.lognormal_lengths(rng, 90, median=14000, sigma=0.5, lo=3000, hi=50000)
        reads = synth.simulate_reads(rng, seqs, lens)
        names = [f"h{h + 1}_read{i:03d}_{int(rng.integers(0, 10**6))}" for i in range(len(reads))]
        synth.write_fasta(str(d / f"r{h}.fq"), names, reads, fastq=True)
        # ---- the file path: tagger, diagonal filter (both steps) through the command lines
        assert kmerpos_annot3.main([str(d / f"r{h}.fq"), str(d / "db"), str(d / "loc"), str(d / f"h{h}.sunkpos")]) == 0
        sunkpos_text = (d / f"h{h}.sunkpos").read_text()
        buf = io.StringIO()
        assert diag_filter_v3.run(str(d / f"h{h}.sunkpos"), str(d / f"hap{h + 1}.fa.fai"), out=buf) == 0
        (d / f"h{h}.diag").write_text(buf.getvalue())
        assert buf.getvalue() == oracle.diag_filter_v3(oracle.parse_sunkpos(sunkpos_text), [n for n, _ in haps[h]])
        buf2 = io.StringIO()
        assert diag_filter_step2.run(str(d / f"h{h}.sunkpos"), str(d / f"h{h}.diag"), out=buf2) == 0
        (d / f"h{h}.diag2").write_text(buf2.getvalue())
        step2_text.append(buf2.getvalue())
        # ---- the device path: passes over device batches, rows appended device to device
        rs = engine.RowSet()
        base, carry = 0, _lib.GS_NONE
        cuts = [0, 17, 18, 60, len(reads)]
        for a, b in zip(cuts, cuts[1:]):
            dev = engine.DeviceReads(engine.ReadBatch.from_ascii([synth.codes_to_ascii(r) for r in reads[a:b]]), 20)
            st, hits, carry, _ = engine.tag(table, dev, carry)
            assert st == 0
            rs.append_pass(table, dev, base, lr)
            base += b - a
            dev.close()
        got = rs.download()
        want = oracle.parse_sunkpos(sunkpos_text)
        assert rs.n == len(want) > 300
        assert [names[i] for i in got["name"].tolist()] == [w[0] for w in want]
        assert got["pos"].tolist() == [w[1] for w in want] and [db.contigs[c] for c in got["contig"].tolist()] == [w[2] for w in want]
        assert got["start"].tolist() == [w[3] for w in want] and got["id"].tolist() == [w[4] for w in want]
        assert np.array_equal(key_contig[got["key"]], got["contig"])
        fai = {n for n, _ in haps[h]}
        seg_name, bc, bg, bd, kept = rs.diag_filter(db.contigs, [1 if c in fai else 0 for c in db.contigs])
        lines = "".join(f"{names[n]}\t{db.contigs[c]}\t{g}\t{chr(dd)}\t{g}\n" for n, c, g, dd in zip(seg_name.tolist(), bc.tolist(), bg.tolist(), bd.tolist()) if c != _lib.GS_NONE)
        assert lines == buf.getvalue() and lines.count("\n") > 40
        rs.keep_filtered()
        got2 = rs.download()
        assert rs.n == kept == buf2.getvalue().count("\n")
        assert "".join(f"{names[n]}\t{p}\t{db.contigs[c]}\t{s}\t{g}\n" for n, p, c, s, g in zip(*(got2[k].tolist() for k in ("name", "pos", "contig", "start", "id")))) == buf2.getvalue()
        sets.append(rs)
        names_of.append(names)
        lens_of.append(lens)
        (d / f"h{h}.rlen").write_text("".join(f"{n}\t{int(l)}\n" for n, l in zip(names, lens)))
    # ---- bad SUNKs over both haplotypes
    assert badsunks.main([str(d / "hap1.fa.fai"), str(d / "hap2.fa.fai"), str(d / "h0.diag2"), str(d / "h1.diag2"), str(d / "bad")]) == 0
    in1 = np.array([1 if c.startswith("h1") else 0 for c in db.contigs], np.uint8)[key_contig]
    in2 = np.array([1 if c.startswith("h2") else 0 for c in db.contigs], np.uint8)[key_contig]
    bad, meancov, d_bad = engine.RowSet.bad_keys(sets[0], sets[1], in1 | (in2 << 1), keep_on_device=True)
    key_id = {}
    for k_, c_, g_ in zip(db.row_loc.tolist(), db.row_contig.tolist(), db.row_group.tolist()):
        key_id[k_] = f"{db.contigs[c_]}:{g_}"
    assert sorted(key_id[int(k_)] for k_ in np.nonzero(bad)[0]) == (d / "bad").read_text().split() and bad.sum() > 3
    # ---- pair stage: all contigs of a haplotype in ONE call == the per-contig command lines
    for h in (0, 1):
        names = names_of[h]
        order = np.argsort(np.array(names, dtype=object), kind="stable")
        rank = np.empty(len(names), np.uint32)
        rank[order] = np.arange(len(names), dtype=np.uint32)
        res = sets[h].pair_stage(rank, (lens_of[h] >= 10000).astype(np.uint8), d_key_bad=d_bad)
        rows = [ln for ln in step2_text[h].split("\n") if ln]
        n_checked = 0
        for ci in res["contigs_in"].tolist():
            cname = db.contigs[ci]
            (d / "c.sunkpos").write_text("".join(r + "\n" for r in rows if r.split("\t")[2] == cname))
            assert process_by_contig.main([str(d / "loc"), str(d / "c.sunkpos"), str(d / f"h{h}.rlen"), str(d / "bad"), str(d / "c.tsv"), str(d / "c.bed")]) == 0
            m = res["row_contig"] == ci
            tsv = "".join(f"{i}\t{names[n]}\n" for i, n in zip(res["row_id"][m].tolist(), res["row_name"][m].tolist()))
            mr = res["region_contig"] == ci
            bed = "".join(f"{cname}\t{s}\t{e}\n" for s, e in zip(res["region_start"][mr].tolist(), res["region_end"][mr].tolist()))
            if tsv:
                assert tsv == (d / "c.tsv").read_text()
                assert bed == (d / "c.bed").read_text()
                n_checked += tsv.count("\n")
            else:
                assert (d / "c.tsv").read_text() == cname + "\n"
            (d / "c.bed").unlink(missing_ok=True)
        assert n_checked > 50 and sorted(set(r.split("\t")[2] for r in rows)) == sorted(db.contigs[c] for c in res["contigs_in"].tolist())
    engine.free_device(d_bad)
    for x in sets:
        x.close()
    lr.close()
    table.close()
    db.close()


def _rows_with_many_contigs(rng, n_contigs_big):
    """Reads whose hits spread over 2 .. n_contigs_big contigs, with ties in (good, hit count) between contigs -- the slot order of
    Nim's Table decides, its capacity grows at the big reads and stays grown for the reads after them."""
    rows = []
    contigs = [f"ctg{j:03d}_{int(rng.integers(0, 10**5))}" for j in range(max(n_contigs_big, 90) + 40)]
    for r in range(60):
        nc = n_contigs_big if r in (20, 41) else (90 if r == 30 else int(rng.integers(2, 9)))
        mine = rng.choice(len(contigs), size=nc, replace=False)
        per = []
        for c in mine:
            nh = int(rng.choice([1, 2, 2, 3, 3, 4]))
            base = int(rng.integers(10**4, 10**6))
            p = np.sort(rng.choice(np.arange(100, 30000), size=nh, replace=False))
            for q in p:
                per.append((int(q), contigs[c], base + int(q) + int(rng.integers(-5, 5)), (base + int(q)) // 1000 * 1000))
        order = rng.permutation(len(per))
        for i in sorted(order.tolist(), key=lambda i: per[i][0]):  # (a read's rows come in pos order)
            q, c, s, g = per[i]
            rows.append((f"read{r}", q, c, s, g))
    return rows, contigs


@pytest.mark.parametrize("big", [44, 200, 300])  # (300 contigs in one read: more than a workgroup's run arrays hold -> global scratch)
def test_table_growth_is_replayed_exactly(tmp_path, big):
    rng = np.random.default_rng(big)
    rows, contigs = _rows_with_many_contigs(rng, big)
    text = "".join(f"{r}\t{p}\t{c}\t{s}\t{g}\n" for r, p, c, s, g in rows)
    (tmp_path / "x.sunkpos").write_text(text)
    fai = [c for i, c in enumerate(contigs) if i % 7 != 3]
    (tmp_path / "x.fai").write_text("".join(f"{c}\t1\n" for c in fai))
    buf = io.StringIO()
    assert diag_filter_v3.run(str(tmp_path / "x.sunkpos"), str(tmp_path / "x.fai"), out=buf) == 0
    exp = oracle.diag_filter_v3(oracle.parse_sunkpos(text), fai)
    assert buf.getvalue() == exp and exp.count("\n") > 30
    # the old host-replay entry point agrees as well (gs_diag_filter: Table order walked on the host)
    sp = engine.SunkPos(str(tmp_path / "x.sunkpos"))
    bc, bg, bd, keep = engine.diag_filter(sp, fai)
    rs = engine.RowSet.from_sunkpos(sp)
    fs = set(fai)
    _, bc2, bg2, bd2, kept = rs.diag_filter(sp.contigs, [1 if c in fs else 0 for c in sp.contigs])
    assert np.array_equal(bc, bc2) and np.array_equal(bg, bg2) and np.array_equal(bd[bc != _lib.GS_NONE], bd2[bc2 != _lib.GS_NONE])
    assert np.array_equal(keep, rs.keep_flags()) and kept == int(keep.sum())
    rs.close()
    sp.close()


def test_pair_stage_prefilters_on_the_device(tmp_path):
    """Lines 50-108 of the script, every branch: bad SUNKs, reads with one distinct ID, duplicate rows (a name that comes back),
    short reads, names that sort differently from file order -- against the oracle's restatement of the script."""
    rng = np.random.default_rng(11)
    rows, rl = [], []
    names = [f"r{int(x):05d}" for x in rng.permutation(300)]
    for i, nm in enumerate(names):
        n = int(rng.choice([1, 2, 3, 6, 12, 30]))
        base = int(rng.integers(10**5, 10**6))
        p = np.sort(rng.choice(np.arange(50, 60000), size=n, replace=False))
        for q in p:
            s = base + int(q) + int(rng.integers(-20, 20)) if rng.random() > 0.1 else int(rng.integers(0, 10**6))
            rows.append(f"{nm}\t{int(q)}\tctgA\t{s}\t{s // 3000 * 3000}\n")
        if rng.random() < 0.15:
            rows.append(rows[-1])  # an exact duplicate
        rl.append(f"{nm}\t{int(rng.choice([500, 9999, 10000, 25000]))}\n")
    rows += rows[5:40]  # names that come back later: new segments, same read for the script's groupby
    text = "".join(rows)
    ids = sorted({int(r.split("\t")[4]) for r in rows})
    bad = "".join(f"ctgA:{i}\n" for i in ids[::9]) + "ctgB:12\nnonsense\nctgA:007\n"
    (tmp_path / "c.sunkpos").write_text(text)
    (tmp_path / "rlen").write_text("".join(rl))
    (tmp_path / "bad").write_text(bad)
    assert process_by_contig.main(["unused.loc", str(tmp_path / "c.sunkpos"), str(tmp_path / "rlen"), str(tmp_path / "bad"), str(tmp_path / "o.tsv"), str(tmp_path / "o.bed")]) == 0
    etsv, ebed = oracle.process_by_contig(text, "".join(rl), bad)
    assert (tmp_path / "o.tsv").read_text() == etsv and etsv.count("\n") > 100
    assert ((tmp_path / "o.bed").read_text() if (tmp_path / "o.bed").exists() else "") == ebed


def test_ids_out_of_range_fail_loudly():
    """Row ids index caller-sized arrays (contig names, keys, name ranks): a count that is too small is an error, not an
    out-of-bounds read on the device."""
    rs = engine.RowSet()
    z = np.zeros(4, np.uint32)
    rs.append_host(np.array([0, 0, 1, 5], np.uint32), z, np.array([0, 1, 2, 2], np.uint32), z, z, np.array([0, 1, 2, 9], np.uint32))
    with pytest.raises(_lib.GsError) as e:
        rs.diag_filter(["a", "b"], [1, 1])  # contig id 2 with two names
    assert e.value.status == _lib.GS_E_ARG
    with pytest.raises(_lib.GsError) as e:
        engine.RowSet.bad_keys(rs, rs, np.full(5, 3, np.uint8))  # key 9 with five keys
    assert e.value.status == _lib.GS_E_ARG
    with pytest.raises(_lib.GsError) as e:
        rs.pair_stage(np.arange(3, dtype=np.uint32), np.ones(3, np.uint8))  # name 5 with three names
    assert e.value.status == _lib.GS_E_ARG
    seg_name, bc, bg, bd, kept = rs.diag_filter(["a", "b", "c"], [1, 1, 1])  # the right count works
    assert len(seg_name) == 3
    rs.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_reads_of_every_kernel_class(tmp_path, seed):
    """gs_diag.hip takes a read by its row count: up to 128 rows four waves to a workgroup, up to 1024 one wave with 32 KiB of LDS,
    longer ones in global scratch.  Reads on both sides of every boundary, one to six contigs each, both orientations, IDs that
    repeat inside the band, exact ties between contigs -- against the oracle and against the host-array implementation."""
    rng = np.random.default_rng(seed)
    contigs = [f"c{j}_{int(rng.integers(0, 999))}" for j in range(9)]
    sizes = [1, 2, 3, 5, 63, 64, 65, 127, 128, 129, 130, 300, 1000, 1023, 1024, 1025, 1500, 2049, 4100] + [int(x) for x in rng.integers(2, 200, 25)]
    rows = []
    for r, n in enumerate(sizes):
        picks = rng.choice(len(contigs), size=int(rng.integers(1, 7)), replace=False)
        weights = rng.dirichlet(np.ones(len(picks)) * 0.6)
        bases = {int(c): int(rng.integers(10**5, 10**7)) for c in picks}
        sign = 1 if rng.random() < 0.5 else -1
        pos = 0
        per = []
        for _ in range(n):
            pos += int(rng.integers(1, 400))
            c = int(rng.choice(picks, p=weights))
            jitter = int(rng.integers(-6000, 6000)) if rng.random() < 0.25 else int(rng.integers(-40, 40))
            start = bases[c] + sign * pos + jitter + (5 * 10**6 if sign < 0 else 0)
            per.append((pos, contigs[c], start, (start // 2000) * 2000))  # coarse IDs: they repeat inside the band
        if r % 5 == 0 and len(picks) >= 2:  # an exact tie: the first contig's rows again on the second (same hit count, same groups' count)
            a, b = contigs[int(picks[0])], contigs[int(picks[1])]
            per = [x for x in per if x[1] != b]
            per += [(p, b, s, g) for p, c_, s, g in per if c_ == a]
            per.sort(key=lambda x: x[0])
        rows += [(f"q{r:03d}", *x) for x in per]
    text = "".join(f"{a}\t{b}\t{c}\t{d}\t{e}\n" for a, b, c, d, e in rows)
    (tmp_path / "x.sunkpos").write_text(text)
    fai = contigs[:8]
    exp = oracle.diag_filter_v3(oracle.parse_sunkpos(text), fai)
    sp = engine.SunkPos(str(tmp_path / "x.sunkpos"))
    rs = engine.RowSet.from_sunkpos(sp)
    fs = set(fai)
    seg_name, bc2, bg2, bd2, kept = rs.diag_filter(sp.contigs, [1 if c in fs else 0 for c in sp.contigs])
    got = "".join(f"{sp.seg_name(s)}\t{sp.contigs[c]}\t{g}\t{chr(d)}\t{g}\n" for s, (c, g, d) in enumerate(zip(bc2.tolist(), bg2.tolist(), bd2.tolist())) if c != _lib.GS_NONE)
    assert got == exp and exp.count("\n") > 25
    bc, bg, bd, keep = engine.diag_filter(sp, fai)
    assert np.array_equal(bc, bc2) and np.array_equal(bg, bg2) and np.array_equal(bd[bc != _lib.GS_NONE], bd2[bc2 != _lib.GS_NONE])
    assert np.array_equal(keep, rs.keep_flags()) and kept == int(keep.sum())
    rs.close()
    sp.close()


def test_more_ties_than_the_first_tie_list_holds():
    """Every read an exact tie between two contigs (two rows, two groups each): the slot order of Nim's Table decides all of them; the
    kernel's list of such reads starts with room for 65 536 and is re-made at the size the first go asked for."""
    n_reads = 70_000
    rng = np.random.default_rng(3)
    contigs = [f"t{j}" for j in range(40)]
    a = rng.integers(0, 40, n_reads)
    b = (a + rng.integers(1, 40, n_reads)) % 40
    base = rng.integers(10**4, 10**6, n_reads)
    name = np.repeat(np.arange(n_reads, dtype=np.uint32), 4)
    pos = np.tile(np.array([100, 200, 1300, 1400], np.uint32), n_reads)
    contig = np.stack([a, b, a, b], 1).reshape(-1).astype(np.uint32)
    start = (np.repeat(base, 4) + pos).astype(np.uint32)
    ident = (start // 1000 * 1000).astype(np.uint32)
    rs = engine.RowSet()
    rs.append_host(name, pos, contig, start, ident, np.zeros(len(name), np.uint32))
    seg_name, bc, bg, bd, kept = rs.diag_filter(contigs, np.ones(40, np.uint8))
    assert np.array_equal(seg_name, np.arange(n_reads)) and (bg == 2).all() and (bd == ord("-")).all()  # ("-" is tried first and "+" is no better)
    # the Table (64 slots, no growth): the contig whose hash lands in the lower slot, the first inserted keeping its slot on a clash
    h = np.array([oracle.murmur3_32(c.encode()) or 314159265 for c in contigs], dtype=np.uint64) & 63
    sa, sb = h[a], h[b]
    sb = np.where(sb == sa, (sb + 1) & 63, sb)
    want = np.where(sa < sb, a, b)
    assert np.array_equal(bc, want) and kept == 2 * n_reads
    rs.close()
