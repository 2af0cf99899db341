#!/usr/bin/env python3
"""Randomised check of the GPU BGZF reader against zlib's bytes, for a number of seconds: the cases of
tests/test_bgzf_gpu.py::test_random_records_levels_strategies_block_sizes with fresh seeds (level, strategy, memory level, member
size x record shapes).  usage: fuzz_bgzf.py [seconds] [first seed]"""
import os
import random
import sys
import tempfile
import time
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import test_bgzf_gpu as T  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
t0 = time.time()
n = 0
with tempfile.TemporaryDirectory(prefix="gs_fuzz_bgzf_", dir="/tmp") as d:
    while time.time() - t0 < seconds:
        rnd = random.Random(seed)
        data = b"".join(T._fuzz_record(rnd, i) for i in range(rnd.randint(20, 500)))
        level = rnd.choice([1, 2, 3, 4, 5, 6, 7, 8, 9])
        strategy = rnd.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY, zlib.Z_FIXED])
        mem_level = rnd.choice([1, 2, 3, 5, 8, 9])
        block = rnd.choice([65280, 65280, 50000, 30000, 4096, 300, 64]) if strategy != zlib.Z_HUFFMAN_ONLY else rnd.choice([30000, 4096, 300])
        path = os.path.join(d, "f.fq.gz")
        with open(path, "wb") as f:
            f.write(T.bgzf(data, level, block, strategy, mem_level=mem_level))
        got = b"".join(T.inflate(path, rnd.choice([1 << 20, 1 << 22, 1 << 26]), chunk_kb=rnd.choice([128, 512, 4096])))
        if got != data:
            print(f"MISMATCH at seed {seed}: level {level}, strategy {strategy}, mem_level {mem_level}, member size {block}, {len(data)} bytes", flush=True)
            sys.exit(1)
        seed += 1
        n += 1
print(f"{n} files equal to zlib's bytes ({time.time() - t0:.0f} s, seeds up to {seed - 1})")
