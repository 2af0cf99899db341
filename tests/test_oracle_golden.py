"""The CPU oracle (oracle/) against the outputs of the reference's own binaries (tests/golden/)."""
import pytest

from conftest import DIAG_CASES, TAG_CASES
from oracle import oracle


@pytest.mark.parametrize("name", TAG_CASES)
def test_tagger_matches_reference_binary(golden, name):
    c = golden(name)
    for run in c.meta["tag"]:
        rc, txt = oracle.tag_files(c.path(run["reads"]), c.path(run["db"]), c.path(run["loc"]))
        assert rc == run["rc"]
        assert txt == c.text(run["out"]), f"{name}: {run['reads']}"


def test_tagger_keyerror_exit_code(golden):
    c = golden("kat2_missing")
    run = c.meta["tag"][0]
    rc, txt = oracle.tag_files(c.path(run["reads"]), c.path(run["db"]), c.path(run["loc"]))
    assert rc == 1 and run["rc"] == 1
    # the reference had flushed nothing or a prefix of the hits when it died
    assert txt.startswith(c.text(run["out"]))


@pytest.mark.parametrize("name", ["kat1", "kat2", "multi_k20", "amy"])
def test_rlen(golden, name):
    c = golden(name)
    for run in c.meta["rlen"]:
        assert oracle.read_lengths(c.path(run["reads"])) == c.text(run["out"])


@pytest.mark.parametrize("name", DIAG_CASES)
def test_diag_filter(golden, name):
    c = golden(name)
    for run in c.meta["diag"]:
        rows = oracle.parse_sunkpos(c.text(run["sunkpos"]))
        fai = [ln.split("\t")[0] for ln in c.text(run["fai"]).splitlines() if ln]
        diag = oracle.diag_filter_v3(rows, fai)
        assert diag == c.text(run["out"])
        assert oracle.diag_filter_step2(c.text(run["sunkpos"]), diag) == c.text(run["out2"])


def test_murmur_known_values():
    # MurmurHash3_x86_32 seed 0 published test vectors
    assert oracle.murmur3_32(b"") == 0
    assert oracle.murmur3_32(b"a") == 0x3C2569B2
    assert oracle.murmur3_32(b"abc") == 0xB3DD93FA
    assert oracle.murmur3_32(b"Hello, world!") == 0xC0363E43
