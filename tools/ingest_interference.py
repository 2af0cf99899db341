#!/usr/bin/env python3
"""Does the copy of the NEXT text block (another thread, its own stream) slow the record parser of the current one down?
GPU box.  Prints parse times alone and with a staging thread running beside them (GAVISUNK_INGEST_TIMING laps on stderr)."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gavisunk_amd import engine  # noqa: E402

rng = np.random.default_rng(1)
n, ln = 20000, 30000
bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * ln, dtype=np.uint8)].tobytes()
text = b"".join(b"@r%d\n" % i + bases[i * ln:(i + 1) * ln] + b"\n+\n" + b"#" * ln + b"\n" for i in range(n))
text2 = bytes(text)
print(f"text {len(text) / 1e9:.2f} GB", flush=True)


def parse_once():
    st = engine.StagedText(text, 0)
    t1 = time.perf_counter()
    dev = engine.DeviceReads.from_staged(st, 20)
    t2 = time.perf_counter()
    st.close()
    dev.close()
    return t2 - t1


for _ in range(3):
    parse_once()
print("alone:", [round(parse_once(), 4) for _ in range(5)], flush=True)
stop = threading.Event()


def stager():
    while not stop.is_set():
        s2 = engine.StagedText(text2, 0)
        s2.close()


th = threading.Thread(target=stager)
th.start()
time.sleep(0.2)
if len(sys.argv) > 1:
    os.environ["GAVISUNK_INGEST_TIMING"] = "1"
print("beside a staging thread:", [round(parse_once(), 4) for _ in range(5)], flush=True)
stop.set()
th.join()
