"""csrc/gs_bgzf.hip without a GPU: the product's source -- the host code that walks bgzip's headers and the kernels that inflate and
check the members -- compiled behind tools/hip_cpu (a stand-in for the HIP runtime: device memory = malloc'ed blocks of the exact
sizes, a block's threads = real threads) and run under AddressSanitizer + UBSan against zlib on files of every block type and on
damaged copies (tools/bgzf_cpu_asan.sh; the GPU pool refuses sanitizer runs).  A slice of it: the script's default is 120 damaged
copies."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="no clang++ (the kernels use ext_vector_type)")
def test_bgzf_kernels_on_the_cpu_under_sanitizers(tmp_path):
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "bgzf_cpu_asan.sh"), str(tmp_path), "16"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    m = re.search(r"bgzf_cpu_asan: (\d+) reads clean \((\d+) of them refused as damaged\)", r.stdout)
    assert m, r.stdout[-2000:]
    assert int(m.group(1)) == 2 * (14 + 1 + 16) and int(m.group(2)) >= 2 + 16  # every file with two block sizes; most damaged copies are refused
    log = (tmp_path / "run.log").read_text()
    assert "MISMATCH" not in log and "ERROR: AddressSanitizer" not in log and "runtime error" not in log
