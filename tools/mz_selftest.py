import ctypes as C
from gavisunk_amd import _lib
L=_lib.lib()
for k,m in [(20,17),(20,16),(20,18),(21,17),(25,21),(20,13),(16,16),(32,28),(31,31)]:
    mis=C.c_uint64(0); bad=(C.c_uint32*4)(); hist=(C.c_uint32*64)()
    rc=L.gs_selftest_minimizer_rule(0,k,m,12345,64,C.byref(mis),bad,hist)
    print(k,m,"rc",rc,"mismatches",mis.value,"of",64*2048,list(bad), "lane",bad[1]&63,"step",bad[1]>>6); print("   ",list(hist))
