#!/bin/bash
# The native gzip reader under AddressSanitizer + UBSan on the CPU (the GPU pool refuses sanitizer runs): files of every block
# type, multi-member, sequence-like and text-like content, several thread counts and read sizes, damaged copies.
set -e
cd "$(dirname "$0")/.."
D=${1:-/tmp/gzasan}
mkdir -p "$D"
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I include -I gavisunk_amd/csrc -o "$D/gz_asan_main" \
    tools/gz_asan_main.cpp gavisunk_amd/csrc/gs_inflate.cpp gavisunk_amd/csrc/gs_host.cpp -lz -pthread -ldl
python3 - "$D" <<'PY'
import gzip, os, sys, zlib
import numpy as np
d = sys.argv[1]
rng = np.random.default_rng(5)
lut = np.frombuffer(b"ACGT", np.uint8)
def fastq(n, lo, hi):
    out = []
    for i in range(n):
        ln = int(rng.integers(lo, hi))
        out.append(b"@r%d\n" % i + lut[rng.integers(0, 4, ln, dtype=np.uint8)].tobytes() + b"\n+\n" + b"#" * ln + b"\n")
    return b"".join(out)
seq = fastq(400, 2000, 60000)
text = b"".join(b"line %d of some text with words %d\n" % (i, int(rng.integers(0, 1000))) for i in range(300000))
rand = rng.integers(0, 256, 3_000_000, dtype=np.uint8).tobytes()
def deflate(data, level, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0):
    c = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
    out = []
    if flush_every:
        for a in range(0, len(data), flush_every):
            out.append(c.compress(data[a:a + flush_every])); out.append(c.flush(zlib.Z_SYNC_FLUSH))
    else:
        out.append(c.compress(data))
    out.append(c.flush())
    return b"".join(out)
files = {
    "seq1.gz": deflate(seq, 1), "seq6.gz": deflate(seq, 6), "seq9.gz": deflate(seq, 9),
    "text6.gz": deflate(text, 6), "fixed.gz": deflate(text[:2_000_000], 6, zlib.Z_FIXED), "huff.gz": deflate(seq[:4_000_000], 6, zlib.Z_HUFFMAN_ONLY),
    "stored.gz": deflate(rand, 0), "rand6.gz": deflate(rand, 6), "sync.gz": deflate(seq[:6_000_000], 1, flush_every=70_000),
    "members.gz": b"".join(gzip.compress(seq[a:a + 1_500_000], 1) for a in range(0, len(seq), 1_500_000)),
    "mixed.gz": gzip.compress(text[:3_000_000], 6) + gzip.compress(rand[:500_000], 0) + gzip.compress(seq[:5_000_000], 1) + gzip.compress(b"", 6),
    "tiny.gz": gzip.compress(b"@r\nACGT\n+\n####\n", 6), "empty.gz": gzip.compress(b"", 6),
    # bgzip's shape: thousands of members of one final block each (member starts are the only piece starts there are)
    "bgzf.gz": b"".join(gzip.compress(seq[a:a + 65280], 6) for a in range(0, 12_000_000, 65280)) + gzip.compress(b"", 6),
}
for k, v in files.items():
    open(os.path.join(d, k), "wb").write(v)
bad = bytearray(files["seq1.gz"]); bad[len(bad) // 2] ^= 0x10
open(os.path.join(d, "bad_flip.gz"), "wb").write(bytes(bad))
open(os.path.join(d, "bad_trunc.gz"), "wb").write(files["seq6.gz"][: len(files["seq6.gz"]) * 2 // 3])
crc = bytearray(files["text6.gz"]); crc[-6] ^= 1
open(os.path.join(d, "bad_crc.gz"), "wb").write(bytes(crc))
print("files:", " ".join(sorted(os.listdir(d))))
PY
export GAVISUNK_GZ_PIECE_KB=256 ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1
for f in seq1 seq6 seq9 text6 fixed huff stored rand6 sync members mixed tiny empty bgzf; do
    for t in 1 3 8; do
        "$D/gz_asan_main" "$D/$f.gz" $t 1000000
    done
    "$D/gz_asan_main" "$D/$f.gz" 4 1000000000
done
unset GAVISUNK_GZ_PIECE_KB
"$D/gz_asan_main" "$D/seq1.gz" 8 100000000
for f in bad_flip bad_trunc bad_crc; do  # damaged files: an error, never a sanitizer report and never wrong bytes passed as right
    if "$D/gz_asan_main" "$D/$f.gz" 4 1000000 2> "$D/$f.err"; then echo "damaged file $f was accepted"; exit 1; fi
    if grep -q "ERROR: AddressSanitizer\|runtime error" "$D/$f.err"; then cat "$D/$f.err"; exit 1; fi
    echo "refused $f: $(tail -1 "$D/$f.err")"
done
# ---- the table parsers (gs_sunkdb_load: threads / single reader, mapping / pread) on files of tools/fuzz_tables.py's generator
python3 - "$D" <<'PY'
import os, sys
sys.path.insert(0, "tools")
import numpy as np
import fuzz_tables
d = sys.argv[1]
for seed in range(1, 13):
    rng = np.random.default_rng(seed)
    k = int(rng.choice([11, 20, 20, 25, 32]))
    big = seed % 2 == 0
    dbt, loct, bad = fuzz_tables.make(rng, k, big)
    open(os.path.join(d, f"t{seed}.db"), "wb").write(dbt)
    open(os.path.join(d, f"t{seed}.loc"), "wb").write(loct)
PY
for seed in 1 2 3 4 5 6 7 8 9 10 11 12; do
    for th in "" 1 3; do
        if [ -n "$th" ]; then export GAVISUNK_LOC_THREADS=$th; else unset GAVISUNK_LOC_THREADS; fi
        "$D/gz_asan_main" --tables "$D/t$seed.db" "$D/t$seed.loc" > "$D/t.out" 2> "$D/t.err" || { cat "$D/t.err"; exit 1; }
        if grep -q "ERROR: AddressSanitizer\|runtime error" "$D/t.err"; then cat "$D/t.err"; exit 1; fi
    done
    tail -1 "$D/t.out"
done
unset GAVISUNK_LOC_THREADS
# ---- rlen's record scanner (gs_rowwriter_rlen) over FASTQ / FASTA texts, truncations and damaged copies of them
python3 - "$D" <<'PY'
import os, sys
import numpy as np
d = sys.argv[1]
rng = np.random.default_rng(11)
lut = np.frombuffer(b"ACGT", np.uint8)
seqs = [lut[rng.integers(0, 4, int(n), dtype=np.uint8)].tobytes() for n in rng.integers(1, 3000, 300)]
fq = b"".join(b"@r%d c\n%s\n+\n%s\n" % (i, s, bytes(rng.integers(33, 74, len(s), dtype=np.uint8))) for i, s in enumerate(seqs))
fa = b"".join(b">c%d\n" % i + b"".join(s[a:a + 60] + b"\n" for a in range(0, len(s), 60)) for i, s in enumerate(seqs))
for name, t in (("fq", fq), ("fq_crlf", fq.replace(b"\n", b"\r\n")), ("fa", fa), ("fa_crlf", fa.replace(b"\n", b"\r\n")), ("nonl", fq[:-1]), ("empty", b""),
                ("fq_small", fq[:fq.index(b"@r5 ")]), ("fa_small", fa[:fa.index(b">c12\n")])):  # (whole records: damaged copies of these are taken or refused byte by byte)
    open(os.path.join(d, "rlen_" + name + ".txt"), "wb").write(t)
PY
for f in fq fq_crlf fa fa_crlf nonl empty fq_small fa_small; do
    "$D/gz_asan_main" --rlen "$D/rlen_$f.txt" 2> "$D/r.err" || { cat "$D/r.err"; exit 1; }
    if grep -q "ERROR: AddressSanitizer\|runtime error" "$D/r.err"; then cat "$D/r.err"; exit 1; fi
done
echo "gz_asan: all clean"
