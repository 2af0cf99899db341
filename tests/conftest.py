import gzip
import json
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class GoldenCase:
    """A directory of tests/golden/ with gzip-stored db/loc expanded into a tmp dir."""

    def __init__(self, name, tmp):
        self.name = name
        self.src = os.path.join(GOLDEN, name)
        self.tmp = str(tmp)
        self.meta = json.load(open(os.path.join(self.src, "case.json")))

    def path(self, f):
        p = os.path.join(self.src, f)
        if os.path.exists(p):
            return p
        gz = p + ".gz"
        if os.path.exists(gz):
            out = os.path.join(self.tmp, f"{self.name}.{f}")
            if not os.path.exists(out):
                with gzip.open(gz, "rb") as a, open(out, "wb") as b:
                    shutil.copyfileobj(a, b)
            return out
        raise FileNotFoundError(p)

    def text(self, f):
        with open(self.path(f)) as fh:
            return fh.read()


@pytest.fixture(scope="session")
def golden(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("golden")
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = GoldenCase(name, tmp)
        return cache[name]

    return get


TAG_CASES = ["kat1", "kat2", "short", "multi_k20", "multi_k21", "multi_k25", "multi_k31", "amy"]
DIAG_CASES = ["kat3", "kat4", "kat5", "tie", "multi_k20", "multi_k21", "multi_k25", "multi_k31", "amy"]
