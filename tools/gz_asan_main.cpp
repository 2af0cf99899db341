// AddressSanitizer / UBSan driver of the native gzip reader (csrc/gs_inflate.cpp), CPU only: inflates a file through gs_gz_* and
// compares the bytes with zlib's gzread.  Built and run by tools/gz_asan.sh (the GPU build cannot run under sanitizers).
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gavisunk.h"

// `--tables db loc`: the two table parsers and the host join (gs_sunkdb_load, gs_sunkdb_keys ...) under the sanitizers
static int tables(const char *db, const char *loc) {
    gs_sunkdb *d = nullptr;
    const int rc = gs_sunkdb_load(db, loc, &d);
    if (rc != GS_OK) {
        printf("refused (%d): %s\n", rc, gs_last_error());
        return 0;
    }
    uint64_t sum = 0;
    const uint64_t nk = gs_sunkdb_n_keys(d), nr = gs_sunkdb_n_rows(d);
    for (uint64_t i = 0; i < nk; i++) sum += gs_sunkdb_keys(d)[i] + gs_sunkdb_key_row(d)[i] + gs_sunkdb_key_loc(d)[i];
    for (uint64_t i = 0; i < nr; i++) sum += gs_sunkdb_row_contig(d)[i] + (uint64_t)gs_sunkdb_row_start(d)[i] + (uint64_t)gs_sunkdb_row_group(d)[i] + gs_sunkdb_row_loc(d)[i];
    printf("ok tables: k %d, %llu keys, %llu rows, %u contigs, checksum %llx\n", gs_sunkdb_k(d), (unsigned long long)nk, (unsigned long long)nr,
           gs_sunkdb_n_contigs(d), (unsigned long long)sum);
    gs_sunkdb_free(d);
    return 0;
}

// `--rlen text`: the record scanner of rlen (gs_rowwriter_rlen) over a text, every one of a few hundred truncations of it and
// damaged copies (a byte replaced by a newline, '@', '>', '+', '\r' or NUL): it may take a block or refuse it, never read outside it.
// The copies are exact-size heap blocks, so that one byte beyond the text is a sanitizer report.
static int rlen_scan(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) return 2;
    std::vector<uint8_t> t;
    uint8_t buf[1 << 16];
    for (size_t n; (n = fread(buf, 1, sizeof buf, f)) > 0;) t.insert(t.end(), buf, buf + n);
    fclose(f);
    gs_rowwriter *w = nullptr;
    if (gs_rowwriter_open("/dev/null", &w) != GS_OK) return 2;
    uint64_t taken = 0, refused = 0, recs = 0, x = 88172645463325252ull;
    auto rnd = [&] {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        return x;
    };
    auto run = [&](const uint8_t *p, size_t n) {
        uint8_t *c = n ? (uint8_t *)malloc(n) : nullptr;
        if (n) memcpy(c, p, n);
        uint64_t r = 0;
        const int rc = gs_rowwriter_rlen(w, c, n, &r);
        free(c);
        if (rc == GS_OK) {
            taken++;
            recs += r;
        } else {
            refused++;
        }
    };
    run(t.data(), t.size());
    for (int i = 0; i < 400 && !t.empty(); i++) run(t.data(), (size_t)(rnd() % (t.size() + 1)));
    const size_t head = t.size() < 20000 ? t.size() : 20000;  // (damaged copies of the head: short, many)
    const uint8_t repl[6] = {'\n', '@', '>', '+', '\r', 0};
    std::vector<uint8_t> d(t.begin(), t.begin() + (long)head);
    for (int i = 0; i < 600 && head; i++) {
        const size_t at = (size_t)(rnd() % head);
        const uint8_t old = d[at];
        d[at] = repl[rnd() % 6];
        run(d.data(), head);
        d[at] = old;
    }
    uint64_t rows = 0;
    gs_rowwriter_close(w, &rows);
    printf("ok rlen %s: %llu blocks taken (%llu records), %llu refused\n", path, (unsigned long long)taken, (unsigned long long)recs, (unsigned long long)refused);
    return 0;
}

int main(int argc, char **argv) {
    if (argc == 4 && !strcmp(argv[1], "--tables")) return tables(argv[2], argv[3]);
    if (argc == 3 && !strcmp(argv[1], "--rlen")) return rlen_scan(argv[2]);
    if (argc < 4) {
        fprintf(stderr, "usage: gz_asan_main file.gz threads want_bytes\n");
        return 2;
    }
    const int threads = atoi(argv[2]);
    const uint64_t want = strtoull(argv[3], nullptr, 10);
    std::vector<uint8_t> ref;
    {
        gzFile g = gzopen(argv[1], "rb");
        if (!g) return 2;
        std::vector<uint8_t> buf(1 << 20);
        int n;
        while ((n = gzread(g, buf.data(), (unsigned)buf.size())) > 0) ref.insert(ref.end(), buf.begin(), buf.begin() + n);
        const bool bad = n < 0;
        gzclose(g);
        if (bad) {
            fprintf(stderr, "zlib refuses %s\n", argv[1]);
            ref.clear();
        }
    }
    gs_gzfile *z = nullptr;
    if (gs_gz_open(argv[1], threads, &z) != GS_OK) {
        fprintf(stderr, "open failed: %s\n", gs_last_error());
        return 1;
    }
    std::vector<uint8_t> got;
    uint64_t nl = 0;
    for (;;) {
        uint8_t *p = nullptr;
        uint64_t n = 0;
        const int rc = gs_gz_read(z, want, nullptr, 0, &p, &n);
        if (rc != GS_OK) {
            fprintf(stderr, "read failed (%d): %s\n", rc, gs_last_error());
            gs_gz_close(z);
            return 3;
        }
        got.insert(got.end(), p, p + n);
        nl += gs_gz_newlines(z);  // (of the stretch just returned)
        gs_gz_free_buffer(p);
        if (!n || gs_gz_at_end(z)) break;
    }
    gs_gz_close(z);
    uint64_t want_nl = 0;
    for (uint8_t c : ref) want_nl += c == '\n';
    if (got.size() != ref.size() || (!ref.empty() && memcmp(got.data(), ref.data(), ref.size()) != 0) || nl != want_nl) {
        fprintf(stderr, "MISMATCH %s: %zu bytes against %zu, newlines %llu against %llu\n", argv[1], got.size(), ref.size(), (unsigned long long)nl,
                (unsigned long long)want_nl);
        return 4;
    }
    printf("ok %s: %zu bytes, %d threads, want %llu\n", argv[1], got.size(), threads, (unsigned long long)want);
    return 0;
}
