"""The prebuilt sidecar of a (k-mer list, location table) pair (gs_sidecar_write / gs_sidecar_open, `kmerpos_annot3 --prepare`):
the reference parses both text files on every run (workflow/src/kmerpos_annot3.nim:20-72) and rule SUNK_annot runs once per
scatter chunk (workflow/rules/tagONT.smk:19-37); with a valid sidecar the drop-in skips parse and build -- and writes the same
bytes, the reference binary's (tests/golden), KeyError included.  A stale sidecar is ignored silently."""
import os
import shutil

import numpy as np
import pytest

from gavisunk_amd import _lib, engine, kmerpos_annot3, synth_gpu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["kat1", "kat2", "amy", "multi_k25", "kat2_missing"])
def test_cli_with_sidecar_writes_the_reference_bytes(golden, name, tmp_path, monkeypatch):
    c = golden(name)
    run = c.meta["tag"][0]
    db, loc = tmp_path / "k.db", tmp_path / "k.loc"
    shutil.copy(c.path(run["db"]), db)
    shutil.copy(c.path(run["loc"]), loc)
    reads, exp = c.path(run["reads"]), c.text(run["out"])

    def cli(out):
        rc = kmerpos_annot3.main([reads, str(db), str(loc), str(tmp_path / out)])
        got = (tmp_path / out).read_text()
        assert rc == run["rc"] and (got == exp if rc == 0 else got.startswith(exp))
        return got

    plain = cli("o1")
    side = engine.sidecar_path(str(loc))
    assert not os.path.exists(side)
    assert kmerpos_annot3.main(["--prepare", str(db), str(loc)]) == 0 and os.path.getsize(side) > 100
    pre = engine.open_sidecar(str(db), str(loc))
    assert pre is not None and pre[0].k == c.meta["k"] and len(pre[0].keys) == 0 and len(pre[0].row_start) > 0
    pre[1].close()
    pre[0].close()
    assert cli("o2") == plain  # through the sidecar
    # the location table changes (one more row): the sidecar is stale and must be ignored, not trusted
    with open(loc, "a") as f:
        f.write(open(loc).readline())
    assert engine.open_sidecar(str(db), str(loc)) is None
    assert cli("o3") == plain  # (a repeated row changes nothing: last duplicate wins, same row content)
    # GAVISUNK_WRITE_SIDECAR=1: an ordinary run refreshes it
    monkeypatch.setenv("GAVISUNK_WRITE_SIDECAR", "1")
    assert cli("o4") == plain
    monkeypatch.delenv("GAVISUNK_WRITE_SIDECAR")
    assert engine.open_sidecar(str(db), str(loc)) is not None
    assert cli("o5") == plain


@pytest.mark.parametrize("layout", ["auto", "minimizer-slots", "compact-hbm", "generic"])
def test_sidecar_keeps_every_layout(layout, golden, tmp_path, monkeypatch):
    """The table travels as it was built (the scalars + array list of a broadcast): whatever layout the first run chose."""
    monkeypatch.setenv("GAVISUNK_TABLE", layout)
    c = golden("amy")
    run = c.meta["tag"][0]
    db, loc = tmp_path / "k.db", tmp_path / "k.loc"
    shutil.copy(c.path(run["db"]), db)
    shutil.copy(c.path(run["loc"]), loc)
    assert kmerpos_annot3.main(["--prepare", str(db), str(loc)]) == 0
    monkeypatch.delenv("GAVISUNK_TABLE")
    pre = engine.open_sidecar(str(db), str(loc))
    assert pre is not None and pre[1].layout.split(":")[0] == {"auto": "minimizer-hbm", "minimizer-slots": "minimizer-hbm"}.get(layout, layout)
    pre[1].close()
    assert kmerpos_annot3.main([c.path(run["reads"]), str(db), str(loc), str(tmp_path / "o")]) == 0
    assert (tmp_path / "o").read_text() == c.text(run["out"])


def test_native_writer_of_the_two_input_files(tmp_path):
    """gs_sunkset_dev_write_files (what bench.py's genome-scale CLI leg feeds the command with) = the row-by-row Python writer."""
    gen = synth_gpu.Genome(300_000, n_contigs=3, seed=9)
    sunks = engine.DeviceSunkSet(gen.ptr, gen.off, 20)
    sunks.write_files(gen.names, str(tmp_path / "db"), str(tmp_path / "loc"))
    contigs = [(gen.names[i], gen.host(int(gen.off[i]), int(gen.off[i + 1]))) for i in range(3)]
    gen.close()
    ref = engine.define_sunks(contigs, 20)
    ref.write_db(str(tmp_path / "db.py"))
    ref.write_loc(str(tmp_path / "loc.py"))
    assert sunks.n == len(ref) > 100_000
    assert (tmp_path / "db").read_bytes() == (tmp_path / "db.py").read_bytes()
    assert (tmp_path / "loc").read_bytes() == (tmp_path / "loc.py").read_bytes()
    sunks.close()
