/*
 * oracle/gs_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's SUNK tagger (GAVISUNK
 * workflow/src/kmerpos_annot3.nim) used only as the parity checker by tests/,
 * __graft_entry__.smoke() and the `cpu_baseline` leg of bench.py.  Nothing in
 * gavisunk_amd/ links, imports or calls it.
 *
 * Parity status: PINNED -- checked byte for byte against outputs of the reference's own
 * prebuilt binary (workflow/scripts/kmerpos_annot3) committed under tests/golden/
 * (tests/test_oracle_golden.py).  The k-mer arithmetic of the reference lives in an
 * un-vendored dependency, brentp/nim-kmer 0.2.6 (`encode`, `slide`; call sites
 * kmerpos_annot3.nim:24,68,88); its published algorithm is restated below.
 *
 * Same algorithm class as the reference on purpose (so that it is an honest CPU baseline):
 * one rolling canonical 2-bit code per window, one hash-set probe per window, one
 * hash-table lookup per raw hit, single thread.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GSO_NONE 0xFFFFFFFFu

/* nim-kmer `encode` table: A/a 0, C/c 1, G/g 2, T/t/U/u 3, EVERY OTHER BYTE 0
 * (kmerpos_annot3.nim:24,68 call sites). */
static uint8_t enc_tab[256];
static int enc_ready = 0;
static void enc_init(void) {
    if (enc_ready) return;
    memset(enc_tab, 0, sizeof enc_tab);
    enc_tab['C'] = enc_tab['c'] = 1;
    enc_tab['G'] = enc_tab['g'] = 2;
    enc_tab['T'] = enc_tab['t'] = enc_tab['U'] = enc_tab['u'] = 3;
    enc_ready = 1;
}

/* `encode`: first base most significant (kmerpos_annot3.nim:24). */
uint64_t gso_encode(const char *s, int k) {
    enc_init();
    uint64_t c = 0;
    for (int i = 0; i < k; i++) c = (c << 2) | enc_tab[(uint8_t)s[i]];
    return c;
}

/* ---------------------------------------------------------------- hash set / map */
typedef struct {
    uint64_t *key;
    uint32_t *val;
    uint64_t mask;
} gso_map;

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

static int map_init(gso_map *m, uint64_t n) {
    uint64_t cap = 16;
    while (cap < 2 * n + 2) cap <<= 1;
    m->key = (uint64_t *)malloc(cap * sizeof(uint64_t));
    m->val = (uint32_t *)malloc(cap * sizeof(uint32_t));
    if (!m->key || !m->val) return -1;
    memset(m->key, 0xFF, cap * sizeof(uint64_t)); /* ~0 is never a canonical code */
    m->mask = cap - 1;
    return 0;
}
static void map_free(gso_map *m) { free(m->key); free(m->val); }
static inline void map_put(gso_map *m, uint64_t k, uint32_t v) { /* later insert overwrites */
    uint64_t h = mix64(k) & m->mask;
    while (m->key[h] != ~0ULL && m->key[h] != k) h = (h + 1) & m->mask;
    m->key[h] = k; m->val[h] = v;
}
static inline int64_t map_get(const gso_map *m, uint64_t k) {
    uint64_t h = mix64(k) & m->mask;
    while (m->key[h] != ~0ULL) {
        if (m->key[h] == k) return (int64_t)m->val[h];
        h = (h + 1) & m->mask;
    }
    return -1;
}

/* ---------------------------------------------------------------- tagger state */
typedef struct {
    int k;
    gso_map set;  /* uniqueKmers: HashSet[uint64]            kmerpos_annot3.nim:20-26 */
    gso_map map;  /* coords: Table[uint64, coord] -> loc row  kmerpos_annot3.nim:53-69 */
    const uint32_t *loc_chrom; /* per loc row */
    const int64_t *loc_group;  /* per loc row */
} gso_table;

gso_table *gso_table_new(int k, uint64_t n_db, const uint64_t *db_codes,
                         uint64_t n_loc, const uint64_t *loc_codes,
                         const uint32_t *loc_chrom, const int64_t *loc_group) {
    gso_table *t = (gso_table *)calloc(1, sizeof *t);
    if (!t) return NULL;
    t->k = k;
    if (map_init(&t->set, n_db) || map_init(&t->map, n_loc)) return NULL;
    for (uint64_t i = 0; i < n_db; i++) map_put(&t->set, db_codes[i], 1);
    /* rows in file order; a later duplicate key overwrites (kmerpos_annot3.nim:68) */
    for (uint64_t i = 0; i < n_loc; i++) map_put(&t->map, loc_codes[i], (uint32_t)i);
    t->loc_chrom = loc_chrom;
    t->loc_group = loc_group;
    return t;
}

void gso_table_free(gso_table *t) {
    if (!t) return;
    map_free(&t->set); map_free(&t->map); free(t);
}

typedef struct { uint32_t read, pos, row; } gso_hit;

typedef struct {
    uint64_t windows, raw_hits, emitted;
} gso_counters;

/*
 * The main loop, kmerpos_annot3.nim:82-96.
 *   prev_row : loc row of the last emitted hit (`prevLoc`), GSO_NONE at the start of a file.
 *              It is NOT reset between reads (declared outside the read loop, :82-84).
 *   `i` (pos) is incremented only when the window was not a suppressed hit: `continue` at :92
 *   skips `inc i` at :96.
 * Returns 0, or -2 when a window is in the db set but absent from the loc map (the reference
 * dies with KeyError, exit code 1, at that point; hits emitted so far are kept), or -3 when
 * `cap` is too small.
 */
int gso_tag(const gso_table *t, const uint8_t *seq, const uint64_t *read_off, uint32_t n_reads,
            uint32_t *prev_row_io, gso_hit *out, uint64_t cap, uint64_t *n_out, gso_counters *ctr) {
    enc_init();
    const int k = t->k;
    const uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
    const int rshift = 2 * (k - 1);
    uint32_t prev = *prev_row_io;
    uint64_t n = 0, windows = 0, raw = 0;
    int rc_ = 0;
    for (uint32_t r = 0; r < n_reads && rc_ == 0; r++) {
        const uint8_t *s = seq + read_off[r];
        const uint64_t len = read_off[r + 1] - read_off[r];
        /* Q6 (observed on the reference binary, tests/golden/short): a read SHORTER than k,
         * the empty read included, still yields exactly ONE window -- its bases followed by
         * zero bytes, which `encode` maps to 'A' -- so the walk runs to max(len, k). */
        const uint64_t end = len < (uint64_t)k ? (uint64_t)k : len;
        uint64_t fwd = 0, rev = 0;
        for (int j = 0; j < k - 1; j++) { /* prime the first k-1 bases */
            uint64_t c = (uint64_t)j < len ? enc_tab[s[j]] : 0;
            fwd = ((fwd << 2) | c) & mask;
            rev = (rev >> 2) | ((3 - c) << rshift);
        }
        uint64_t i = 0; /* the reference's `i` */
        for (uint64_t e = k - 1; e < end; e++) {
            uint64_t c = e < len ? enc_tab[s[e]] : 0;
            fwd = ((fwd << 2) | c) & mask;
            rev = (rev >> 2) | ((3 - c) << rshift);
            const uint64_t canon = fwd < rev ? fwd : rev; /* slide(): x[0] = min(fwd, rc) */
            windows++;
            if (map_get(&t->set, canon) >= 0) {          /* :89 `x[0] in uniqueKmers` */
                raw++;
                const int64_t row = map_get(&t->map, canon); /* :90 coords[x[0]] */
                if (row < 0) { rc_ = -2; break; }            /* KeyError */
                if (prev != GSO_NONE && t->loc_chrom[row] == t->loc_chrom[prev] &&
                    t->loc_group[row] == t->loc_group[prev])
                    continue;                              /* :92, and no `inc i` */
                if (n >= cap) { rc_ = -3; break; }
                out[n].read = r; out[n].pos = (uint32_t)i; out[n].row = (uint32_t)row; n++;
                prev = (uint32_t)row;
            }
            i++;
        }
    }
    *prev_row_io = prev;
    *n_out = n;
    if (ctr) { ctr->windows = windows; ctr->raw_hits = raw; ctr->emitted = n; }
    return rc_;
}

/* Canonical code of every window of one sequence (test helper: checks the product's packer
 * and rolling code against the restatement). */
void gso_canonical_windows(const uint8_t *s, uint64_t len, int k, uint64_t *out) {
    enc_init();
    const uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
    const int rshift = 2 * (k - 1);
    uint64_t fwd = 0, rev = 0;
    for (uint64_t e = 0; e < len; e++) {
        uint64_t c = enc_tab[s[e]];
        fwd = ((fwd << 2) | c) & mask;
        rev = (rev >> 2) | ((3 - c) << rshift);
        if (e + 1 >= (uint64_t)k) out[e + 1 - k] = fwd < rev ? fwd : rev;
    }
}
