"""csrc/gs_ingest.hip without a GPU: the device record parser -- staging code and kernels, the product's source -- compiled behind
tools/hip_cpu and run under AddressSanitizer + UBSan (tools/ingest_cpu_asan.sh): regular texts equal the host record reader's reads,
irregular ones are refused, randomly damaged copies do neither harm.  A slice: the script's default is 60 mutated copies."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="no clang++")
def test_record_parser_on_the_cpu_under_sanitizers(tmp_path):
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "ingest_cpu_asan.sh"), str(tmp_path), "6"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    m = re.search(r"ingest_cpu_main: (\d+) taken, (\d+) refused, 0 bad", r.stdout)
    assert m, r.stdout[-2000:]
    assert int(m.group(1)) >= 13 and int(m.group(1)) + int(m.group(2)) == 20 + 6
    log = (tmp_path / "run.log").read_text()
    assert "MISMATCH" not in log and "ERROR: AddressSanitizer" not in log and "runtime error" not in log
