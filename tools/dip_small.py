import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from gavisunk_amd import engine, synth_gpu, _lib
import ctypes as C
os.environ["GAVISUNK_DEBUG_BUILD"] = "1"
for kind in ("diploid", "config3"):
    w = dict(synth_gpu.WORKLOADS[kind])
    gen = synth_gpu.make_genome(w, 100.0)
    sunks = engine.DeviceSunkSet(gen.ptr, gen.off, 20)
    print(kind, "keys", sunks.n, "groups", sunks.n_groups, flush=True)
    t = sunks.table()
    print(t.layout, flush=True)
    t.close(); sunks.close(); gen.close()
