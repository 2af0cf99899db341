import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from gavisunk_amd import engine, synth, _lib
import ctypes as C
# a config-2-like reads file: 10 000 reads, ~250 Mbase, one line per read
rng = np.random.default_rng(1)
path = "/tmp/reads.fa"
with open(path, "wb") as f:
    for i in range(10000):
        n = int(rng.integers(5000, 45000))
        f.write(b">read%d\n" % i + synth.codes_to_ascii(rng.integers(0, 4, n, dtype=np.uint8)) + b"\n")
print("file MB", os.path.getsize(path) / 1e6)
L = _lib.lib()
for rep in range(3):
    t0 = time.perf_counter()
    blocks = list(engine.text_blocks(path, 1 << 30))
    t1 = time.perf_counter()
    d = engine.DeviceReads.from_text(blocks[0], 20)
    t2 = time.perf_counter()
    print(f"text_blocks {t1-t0:.3f} s, from_text {t2-t1:.3f} s, reads {d.n_reads} bases {d.n_bases}")
    d.close()
    t0 = time.perf_counter()
    hb = list(engine.read_batches(path))
    t1 = time.perf_counter()
    dd = [engine.DeviceReads(b, 20) for b in hb]
    t2 = time.perf_counter()
    print(f"host parse {t1-t0:.3f} s, upload {t2-t1:.3f} s")
    for x in dd: x.close()
    for b in hb: b.close()
