"""Files for tools/bgzf_cpu_asan.sh: bgzip-shaped `.gz` of every block type over records built to reach every path of the decoder, and
damaged copies (one flipped bit, twenty, a cut, a BSIZE or ISIZE that lies).  usage: bgzf_cpu_files.py <dir> <damaged copies>"""
import os, random, struct, sys, zlib
d = sys.argv[1]
ACGT = bytes(b"ACGT"[i & 3] for i in range(256))
def bgzf(data, level=6, block=65280, strategy=zlib.Z_DEFAULT_STRATEGY, eof=True, mem_level=8):
    out = []
    for a in list(range(0, len(data), block)) + ([len(data)] if eof else []):
        chunk = data[a:a + block] if a < len(data) else b""
        c = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
        raw = c.compress(chunk) + c.flush()
        bsize = 12 + 6 + len(raw) + 8
        assert bsize <= 65536
        out.append(b"\x1f\x8b\x08\x04" + b"\0" * 6 + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + raw +
                   struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    return b"".join(out)
rnd = random.Random(11)
def rec(i):
    n = rnd.choice([1, 2, 7, 40, 300, 2000, 9000, 30000])
    kind = rnd.randrange(5)
    if kind == 0: seq = rnd.randbytes(n).translate(ACGT)
    elif kind == 1:
        p = rnd.randbytes(rnd.randint(1, 20)).translate(ACGT); seq = (p * (n // len(p) + 1))[:n]
    elif kind == 2:
        u = rnd.randbytes(rnd.randint(33, 400)).translate(ACGT); seq = (u * (n // len(u) + 1))[:n]
    elif kind == 3: seq = b"A" * n
    else: seq = bytes(rnd.choice(b"ACGTN") for _ in range(n))
    q = rnd.randrange(3)
    qual = b"#" * n if q == 0 else bytes(33 + rnd.randrange(94) for _ in range(n)) if q == 1 else bytes(33 + (j * 7) % 40 for j in range(n))
    return b"@r%d x\n%s\n+\n%s\n" % (i, seq, qual)
data = b"".join(rec(i) for i in range(500))
files = {
    "l1": bgzf(data, 1), "l6": bgzf(data, 6), "l9": bgzf(data, 9), "stored": bgzf(data[:2_000_000], 0, 60000), "fixed": bgzf(data, 6, strategy=zlib.Z_FIXED),
    "huff": bgzf(data[:3_000_000], 6, 40000, zlib.Z_HUFFMAN_ONLY), "small": bgzf(data[:600_000], 6, 777), "rle": bgzf(data, 6, strategy=zlib.Z_RLE),
    "mem1": bgzf(data, 6, mem_level=1), "noeof": bgzf(data[:1_000_000], 6, eof=False), "tiny": bgzf(b"@a\nA\n+\n#\n", 6), "empty": bgzf(b"", 6),
    "fa": bgzf(b"".join(b">c%d\n%s\n" % (i, rnd.randbytes(rnd.randint(1, 50000)).translate(ACGT)) for i in range(100)), 6),
}
def member(chunk, level, strategy=zlib.Z_DEFAULT_STRATEGY):
    return bgzf(chunk, level, strategy=strategy, eof=False) if chunk else bgzf(b"", level, strategy=strategy)
# empty members that are NOT bgzip's end-of-file marker (a stored block of length 0, an empty dynamic-less block behind a sync flush)
# between members that carry text: decoded like any other member, nothing written
def empty_sync():
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = c.flush(zlib.Z_SYNC_FLUSH) + c.flush()
    return (b"\x1f\x8b\x08\x04" + b"\0" * 6 + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(raw) + 8 - 1) + raw + struct.pack("<II", 0, 0))
files["empties"] = b"".join(member(data[a:a + 50000], 6) + (member(b"", 0) if (a // 50000) % 3 == 0 else empty_sync() if (a // 50000) % 3 == 1 else member(b"", 6))
                            for a in range(0, 1_500_000, 50000)) + member(b"", 0)
for k, v in files.items(): open(os.path.join(d, k + ".gz"), "wb").write(v)
# a member full of text whose ISIZE field reads 0 (what the end-of-file marker's does): zlib fails its size check; so must this reader
b = bytearray(base := files["l6"])
at = 0
for _ in range(5): at += struct.unpack_from("<H", b, at + 16)[0] + 1
struct.pack_into("<I", b, at + struct.unpack_from("<H", b, at + 16)[0] + 1 - 4, 0)
open(os.path.join(d, "bad_isize0.gz"), "wb").write(bytes(b))
# damaged copies: flipped bits all over (headers, deflate data, trailers), cuts
for j in range(int(sys.argv[2])):
    b = bytearray(base if j % 3 else files["l1"])
    m = j % 4
    if m == 0:
        b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
    elif m == 1:
        for _ in range(20): b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
    elif m == 2:
        b = b[:rnd.randrange(30, len(b))]
    else:  # a lying header: BSIZE or ISIZE of some member
        at = 0; k = rnd.randrange(40)
        for _ in range(k):
            at += struct.unpack_from("<H", b, at + 16)[0] + 1
            if at + 18 > len(b): at = 0; break
        if rnd.randrange(2): struct.pack_into("<H", b, at + 16, rnd.randrange(65536))
        else:
            bs = struct.unpack_from("<H", b, at + 16)[0] + 1
            struct.pack_into("<I", b, at + bs - 4, rnd.choice([0, 1, 65535, 65536, 70000, 2**32 - 1, rnd.randrange(65536)]))
    open(os.path.join(d, "bad%03d.gz" % j), "wb").write(bytes(b))
