/*
 * gavisunk.h -- C-ABI of the MI355X-native SUNK-tagging engine (libgavisunk.so).
 *
 * Drop-in boundary.  GAVISUNK has no FFI: its hot path is four single-threaded Nim programs
 * invoked by Snakemake shell lines (workflow/rules/tagONT.smk:36,73,92,110) plus one Python
 * script (tagONT.smk:189).  The entry points below are what a binding for that path has to
 * reach; every one cites the reference code it replaces.  The Python host stubs in
 * gavisunk_amd/ (same positional argv as the reference programs) call them through ctypes;
 * INTEGRATION.md shows the one-line change per Snakemake rule.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every function returns 0 or a negative
 *     gs_status, and gs_last_error() (thread-local) explains the last failure;
 *   - the caller owns every host buffer it passes; the library owns device memory behind the
 *     opaque handles; one handle lives on one GPU; a handle is not thread-safe, distinct
 *     handles may be used from distinct threads;
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point
 *     fails with GS_E_NODEVICE.
 */
#ifndef GAVISUNK_H
#define GAVISUNK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gs_status {
    GS_OK = 0,
    GS_E_ARG = -1,       /* bad argument */
    GS_E_KEYERROR = -2,  /* a read hit a k-mer that is in the db but has no loc row: the
                            reference dies with KeyError, exit 1 (kmerpos_annot3.nim:90) */
    GS_E_CAPACITY = -3,  /* output buffer too small; *n_out holds the size needed */
    GS_E_NODEVICE = -4,  /* no HIP device / HIP runtime error */
    GS_E_IO = -5,        /* file cannot be opened or parsed (kmerpos_annot3.nim:59-60) */
    GS_E_NOMEM = -6,
    GS_E_IRREGULAR = -7 /* gs_reads_from_text: the text is outside the regular FASTA / FASTQ subset it parses: use gs_fastx_* */
} gs_status;

#define GS_NONE 0xFFFFFFFFu    /* "no previous location" (prevLoc before the first hit) */
#define GS_MISSING 0xFFFFFFFEu /* loc id of a db k-mer without a loc row */

const char *gs_last_error(void);
const char *gs_version(void);
int gs_device_count(void);
/* hipDeviceSynchronize on `device`: everything the library (or RCCL through gs_comm_*) enqueued there has finished */
int gs_device_synchronize(int device);
/* on != 0: this process waits for the device by blocking instead of spinning (hipDeviceScheduleBlockingSync): for command lines whose
 * host threads need every CPU (the .fq.gz reader); call before anything else touches the device.  The engine's default is to spin. */
int gs_device_blocking_sync(int device, int on);
/* Self-test of the minimiser-addressed table's window -> region rule (gs_mz.h): on `n_tiles` tiles of 2048 windows of
 * pseudo-random and low-complexity sequence (from `seed`), the wave-parallel sliding form the tagging kernel uses is compared
 * with the rule applied window by window (the form the table build uses).  *mismatches = windows that differ (0 = pass);
 * first_bad = {tile, window, got, expected} of one of them; lane_hist (may be NULL) = mismatches by wavefront lane.
 * No reference counterpart: the table layout is this library's. */
int gs_selftest_minimizer_rule(int device, int k, int m, uint64_t seed, uint32_t n_tiles, uint64_t *mismatches, uint32_t first_bad[4],
                               uint32_t lane_hist[64]);

/* ------------------------------------------------------------------------------------------
 * Host-side parsing (native, like the reference's Nim programs).  No GPU needed.
 * ---------------------------------------------------------------------------------------- */

/* The SUNK list + location table as the reference loads them:
 *   db  : one k-mer per line, k = length of line 1          (kmerpos_annot3.nim:20-38)
 *   loc : TSV chrom, start, kmer, group; a later row with the same k-mer overwrites an
 *         earlier one                                        (kmerpos_annot3.nim:53-69)
 * Both plain or gzip.  `encode` semantics: A/a 0, C/c 1, G/g 2, T/t/U/u 3, any other byte 0,
 * first base most significant (nim-kmer 0.2.6 `encode`, call sites kmerpos_annot3.nim:24,68). */
typedef struct gs_sunkdb gs_sunkdb;
int gs_sunkdb_load(const char *db_path, const char *loc_path, gs_sunkdb **out);
void gs_sunkdb_free(gs_sunkdb *db);
int gs_sunkdb_k(const gs_sunkdb *db);
/* Keys of the tagger's table = distinct db k-mers.  Arrays are valid until gs_sunkdb_free. */
uint64_t gs_sunkdb_n_keys(const gs_sunkdb *db);
const uint64_t *gs_sunkdb_keys(const gs_sunkdb *db);     /* code of each key (as in the db; canonical there) */
const uint32_t *gs_sunkdb_key_row(const gs_sunkdb *db);  /* winning loc row of the key, GS_MISSING if none */
const uint32_t *gs_sunkdb_key_loc(const gs_sunkdb *db);  /* dense id of that row's (chrom, group), GS_MISSING if none */
/* loc rows (file order) */
uint64_t gs_sunkdb_n_rows(const gs_sunkdb *db);
const uint32_t *gs_sunkdb_row_contig(const gs_sunkdb *db);
const int64_t *gs_sunkdb_row_start(const gs_sunkdb *db);
const int64_t *gs_sunkdb_row_group(const gs_sunkdb *db);
const uint32_t *gs_sunkdb_row_loc(const gs_sunkdb *db);  /* dense (chrom, group) id per row */
uint32_t gs_sunkdb_n_contigs(const gs_sunkdb *db);
const char *gs_sunkdb_contig_name(const gs_sunkdb *db, uint32_t contig);

/* FASTA/FASTQ reader with kseq semantics, plain or gzip, multi-line records; name = header up
 * to the first whitespace (the reference iterates `readfq(reads)`, kmerpos_annot3.nim:85,
 * rlen.nim:13).  A batch holds whole records, 2-bit packed for the device:
 *   words    : uint64, 32 bases per word, first base in the two most significant bits; each
 *              read starts on a word boundary and is followed by >= 1 zero word
 *   word_off : n_reads+1 offsets into `words`;  len : bases per read (what `rlen` prints). */
typedef struct gs_fastx gs_fastx;
typedef struct gs_batch gs_batch;
int gs_fastx_open(const char *path, gs_fastx **out);
void gs_fastx_close(gs_fastx *fx);
/* Reads records until `max_bases` bases or `max_reads` records are in the batch (at least
 * one record).  *out = NULL at end of file. */
int gs_fastx_next_batch(gs_fastx *fx, uint64_t max_bases, uint32_t max_reads, gs_batch **out);
void gs_batch_free(gs_batch *b);
uint32_t gs_batch_n_reads(const gs_batch *b);
uint64_t gs_batch_n_bases(const gs_batch *b);
const uint64_t *gs_batch_words(const gs_batch *b);
uint64_t gs_batch_n_words(const gs_batch *b);
const uint64_t *gs_batch_word_off(const gs_batch *b);
const uint32_t *gs_batch_len(const gs_batch *b);
const char *gs_batch_name(const gs_batch *b, uint32_t read);
/* all names at once: NUL-terminated, back to back from *base; name r starts at (*off)[r]; *bytes = length of the whole blob */
int gs_batch_names(const gs_batch *b, const char **base, const uint64_t **off, uint64_t *bytes);
/* A batch from memory: ASCII bases of all reads back to back, off[n_reads+1] in bases.
 * Packing applies the `encode` table above. */
int gs_batch_from_ascii(const uint8_t *seq, const uint64_t *off, uint32_t n_reads, gs_batch **out);

/* Parallel gzip inflation (csrc/gs_inflate.cpp) for the `.fq.gz` files rule SUNK_annot is given (tagONT.smk:17,36; the
 * reference reads them through zlib inside `readfq`, kmerpos_annot3.nim:85).  Workers find deflate block starts in their piece
 * of the compressed file, decode with the unknown 32 KiB window kept symbolic, and the pieces are chained from the true start of
 * the stream (only proven block boundaries are used); CRC-32 and size of every member are checked.  Any gzip file works
 * (multi-member, stored / fixed blocks: those decode on one thread); a damaged file fails with GS_E_IO where zlib fails.
 *   gs_gz_open    threads = 0: one per hardware thread (at most 32; GAVISUNK_GZ_THREADS overrides)
 *   gs_gz_read    `prefix` (what the caller carried over from the stretch before: an unfinished record), then at least `want`
 *                 new bytes unless the file ends first.  *data is handed over: free it with gs_gz_free_buffer.
 *   gs_text_record_cut   where a block of FASTA / four-line FASTQ text can be cut so that it holds whole records */
typedef struct gs_gzfile gs_gzfile;
int gs_gz_open(const char *path, int threads, gs_gzfile **out);
int gs_gz_read(gs_gzfile *z, uint64_t want, const uint8_t *prefix, uint64_t prefix_len, uint8_t **data, uint64_t *n);
void gs_gz_free_buffer(uint8_t *p);
int gs_gz_at_end(const gs_gzfile *z);
const char *gs_gz_mode(const gs_gzfile *z);
uint64_t gs_gz_pieces(const gs_gzfile *z, uint64_t *dropped);
void gs_gz_close(gs_gzfile *z);
uint64_t gs_text_record_cut(const uint8_t *text, uint64_t n);
/* '\n' bytes among the new bytes (the prefix not counted) of the stretch gs_gz_read returned last: counted while they are written */
uint64_t gs_gz_newlines(const gs_gzfile *z);
/* gs_text_record_cut when the number of '\n' in the text is known already (no pass over the text) */
uint64_t gs_text_record_cut_counted(const uint8_t *text, uint64_t n, uint64_t n_newlines);

/* ------------------------------------------------------------------------------------------
 * Device side.
 * ---------------------------------------------------------------------------------------- */

/* Device-resident SUNK table: open-addressed hash, built on the GPU.  Replaces
 * `uniqueKmers: HashSet[uint64]` + `coords: Table[uint64, coord]`
 * (kmerpos_annot3.nim:20-26, 44-69).
 *   keys[n]    distinct canonical codes (2 bits/base, k <= 32)
 *   loc[n]     dense id of the key's (chrom, group) -- what `curLoc == prevLoc` compares
 *              (kmerpos_annot3.nim:91-92) -- or GS_MISSING
 *   payload[n] what a hit reports for this key (the host stubs pass the loc row) */
typedef struct gs_table gs_table;
int gs_table_create(int device, int k, uint64_t n, const uint64_t *keys, const uint32_t *loc,
                    const uint32_t *payload, gs_table **out);
/* The same from arrays that already are in DEVICE memory of `device` (e.g. the result of gs_sunks_define_device: no
 * text or host round trip between SUNK definition and tagging).  d_payload may be NULL:
 * payload[i] = i.  The arrays are read during the call only; the caller keeps owning them. */
int gs_table_create_device(int device, int k, uint64_t n, const void *d_keys, const void *d_loc, const void *d_payload,
                           gs_table **out);
/* The table of a parsed <kmers.db> + <kmer.loc> pair (gs_sunkdb_load), the join made on the device: distinct db k-mers (the
 * reference's HashSet, kmerpos_annot3.nim:20-38), each with the LAST loc row that carries it (:68) -- two radix sorts and a merge
 * instead of the reference's per-run hash-map inserts.  A db k-mer without loc row keeps loc = GS_MISSING (KeyError when a read
 * meets it, :90). */
int gs_table_create_from_db(int device, const gs_sunkdb *db, gs_table **out);
void gs_table_destroy(gs_table *t);
uint64_t gs_table_bytes(const gs_table *t);
/* Human-readable description of the layout chosen for this table: "minimizer-hbm" (the default for k >= 16: 64-byte regions
 * addressed by a mod-minimiser of the k-mer, so that consecutive windows share one memory request -- the fastest layout on
 * every table size), "compact-lds" / "compact-l2" (small tables: an exact compact copy probed from LDS / the
 * XCD-partitioned L2s), "compact-hbm", "generic" (k < 16, db k-mers without loc row).
 * GAVISUNK_TABLE=generic|compact-lds|compact-l2|compact-hbm|minimizer (read once, when the table is created) forces one. */
const char *gs_table_layout(const gs_table *t);

/* Device-resident batch of packed reads (+ the per-tile index the kernels need). */
typedef struct gs_reads gs_reads;
int gs_reads_upload(int device, const uint64_t *words, uint64_t n_words, const uint64_t *word_off,
                    const uint32_t *len, uint32_t n_reads, int k, gs_reads **out);
/* Scatter-chunk emulation.  GAVISUNK's workflow cuts a haplotype's reads into `nchunks` files
 * (`seqtk seq -F '#' | rustybam fastq-split`, tagONT.smk:2-18), runs the tagger once per file -- `prevLoc` starts empty in each
 * (tagONT.smk:19-37) -- and concatenates the outputs in scatter order (tagONT.smk:112-131).  One pass can reproduce that:
 *   gs_reads_upload_ordered   the batch's reads are tagged in `order` (tagging position i = batch read order[i]: chunk 0's reads
 *                             first, then chunk 1's, ...); gs_hit.read is the tagging position
 *   gs_reads_set_seeds        before the read at tagging position seed_at[j] the carried location is FORCED to seed[j]: GS_NONE at
 *                             the first read of a chunk, or what that chunk carried out of the previous device batch
 *   gs_tag_segment_carries    after a pass: out[j] = the location carried up to seed_at[j] (by the reads of the segment before
 *                             it); the last segment's is the pass's carry_out */
int gs_reads_upload_ordered(int device, const uint64_t *words, uint64_t n_words, const uint64_t *word_off, const uint32_t *len,
                            uint32_t n_reads, const uint32_t *order, uint32_t n_order, int k, gs_reads **out);
int gs_reads_set_seeds(gs_reads *r, const uint32_t *seed_at, const uint32_t *seed, uint32_t n);
int gs_tag_segment_carries(gs_table *t, gs_reads *r, uint32_t *out);
/* The same from unpacked reads: ASCII bases back to back, off[n_reads+1] in bytes; the 2-bit packing (with the
 * `encode` table above) runs on the GPU. */
int gs_reads_upload_ascii(int device, const uint8_t *seq, const uint64_t *off, uint32_t n_reads, int k, gs_reads **out);
/* The same from bases that already are in DEVICE memory of `device` (d_seq: ASCII back to back; off: host offsets).  The
 * bases are read during the call only. */
int gs_reads_from_device_ascii(int device, const void *d_seq, const uint64_t *off, uint32_t n_reads, int k, gs_reads **out);
/* The same from the TEXT of a FASTA / FASTQ block (whole records, decompressed, below 4 GiB): records are parsed on the GPU
 *  -- line starts, record boundaries, names, lengths -- and the bases gathered and packed there.
 * Replaces `for rec in readfq(reads)` (kmerpos_annot3.nim:85) + `encode` for texts in the regular subset (one-line or
 * multi-line FASTA; four-line FASTQ; see csrc/gs_ingest.hip); any other text returns GS_E_IRREGULAR and is read with
 * gs_fastx_* instead, which gives the same reads for the texts both accept.  The index (host arrays, one entry per read in
 * file order) says where each read's name is in `text` and how many bases it has. */
typedef struct gs_textindex gs_textindex;
int gs_reads_from_text(int device, const uint8_t *text, uint64_t n_bytes, int k, gs_reads **out, gs_textindex **index);
/* The two halves of gs_reads_from_text, for callers that overlap them with other work: gs_text_stage copies the text to the
 * device (through pinned buffers, on a stream of its own; safe to call from a second thread while the first loads the SUNK
 * table), gs_reads_from_staged_text parses what was staged.  `text` must stay readable for the names until the index is used. */
typedef struct gs_textstage gs_textstage;
int gs_text_stage(int device, const uint8_t *text, uint64_t n_bytes, gs_textstage **out);
void gs_text_stage_free(gs_textstage *st);
int gs_reads_from_staged_text(const gs_textstage *stage, int k, gs_reads **out, gs_textindex **index);

/* DEFLATE on the GPU for BGZF-shaped `.fq.gz` (csrc/gs_bgzf.hip): what `rustybam fastq-split` and bgzip write, and what rule
 * SUNK_annot is given (tagONT.smk:17,36) -- a gzip file of members of at most 64 KiB of text, each with its compressed size in the
 * header and its inflated size in the trailer.  The host walks the headers and hands the COMPRESSED bytes to the device; a lane per
 * member inflates straight to the member's place in a text block (CRC-32 and size of every member checked: a damaged file fails
 * with GS_E_IO where zlib fails); the block -- whole records, the rest carried to the next -- goes to gs_reads_from_staged_text
 * where it lies.  gs_bgzf_open: GS_E_IRREGULAR = not a BGZF file (read it with gs_gz_*).  gs_bgzf_next: *stage = NULL at the end
 * of the file, or when not one whole record fit (ask again with a larger `want`); GS_E_IRREGULAR = a member that is not in the
 * container (a bgzip file with plain gzip members appended: read the file with gs_gz_*, from the start). */
typedef struct gs_bgzf gs_bgzf;
int gs_bgzf_open(const char *path, int device, gs_bgzf **out);
int gs_bgzf_next(gs_bgzf *z, uint64_t want_text_bytes, gs_textstage **stage);
int gs_bgzf_at_end(const gs_bgzf *z);
uint64_t gs_bgzf_bytes_in(const gs_bgzf *z);
void gs_bgzf_close(gs_bgzf *z);
/* the names of a parsed block whose text exists only on the device: blob_off[n + 1] (host) <- running sums of name_len, blob <- the
 * names back to back (gathered on the device, one copy); blob may be NULL to learn the size (blob_off[n]) */
int gs_textstage_names(const gs_textstage *stage, const uint64_t *name_off, const uint32_t *name_len, uint32_t n, uint8_t *blob, uint64_t *blob_off);
/* (tests, tools) a staged text on the host: *n <- its size; out may be NULL */
int gs_textstage_copy(const gs_textstage *stage, uint8_t *out, uint64_t cap, uint64_t *n);
void gs_textindex_free(gs_textindex *ix);
uint32_t gs_textindex_n_reads(const gs_textindex *ix);
const uint64_t *gs_textindex_name_off(const gs_textindex *ix);
const uint32_t *gs_textindex_name_len(const gs_textindex *ix);
const uint32_t *gs_textindex_len(const gs_textindex *ix);
void gs_reads_destroy(gs_reads *r);
uint64_t gs_reads_n_bases(const gs_reads *r);
uint64_t gs_reads_n_windows(const gs_reads *r);

/* One emitted hit = one row of the reference's output `name \t i \t chrom \t start \t group`
 * (kmerpos_annot3.nim:93): read = index of the read in the batch, pos = the reference's `i`
 * (window index minus the suppressed hits before it in the same read, kmerpos_annot3.nim:92,96),
 * payload = the key's payload (loc row). */
typedef struct gs_hit {
    uint32_t read;
    uint32_t pos;
    uint32_t payload;
} gs_hit;

typedef struct gs_counters {
    uint64_t windows;      /* windows examined (= hash probes) */
    uint64_t raw_hits;     /* windows found in the SUNK set */
    uint64_t emitted;      /* rows of output */
    uint64_t tiles;        /* 2048-window tiles */
    uint64_t launches;     /* tag passes accumulated in the two times below */
    double probe_ms;       /* HIP-event time of the probe kernel, summed over `launches` */
    double total_ms;       /* HIP-event time of the whole pass (all kernels), summed */
    uint64_t fallback_tiles; /* tiles of the last pass that the fast probe path handed to its exact fallback */
} gs_counters;

/* The tagger's main loop (kmerpos_annot3.nim:82-96) over a resident batch.
 *   carry_in  : loc id of the last raw hit before this batch in the same input file
 *               (`prevLoc` is never reset between reads), GS_NONE at the start of a file;
 *   carry_out : the same for the next batch.
 * gs_tag = gs_tag_launch + gs_tag_finish.  gs_tag_launch only enqueues the kernels on the
 * table's stream; gs_tag_finish waits for them and copies the hit list out (`out` may be
 * NULL to just count).  On GS_E_KEYERROR, *n_out = hits emitted before the offending window. */
int gs_tag(gs_table *t, gs_reads *r, uint32_t carry_in, gs_hit *out, uint64_t cap, uint64_t *n_out,
           uint32_t *carry_out, gs_counters *ctr);
int gs_tag_launch(gs_table *t, gs_reads *r, uint32_t carry_in);
int gs_tag_finish(gs_table *t, gs_reads *r, gs_hit *out, uint64_t cap, uint64_t *n_out,
                  uint32_t *carry_out, gs_counters *ctr);
/* Split pass for read-sharded multi-GPU runs (the shard's carry-in is not known until the
 * summaries of the shards before it have been exchanged): phase 1 probes and reports whether
 * the shard has a raw hit and the loc id of its last one; phase 2 takes the carry-in. */
int gs_tag_phase1(gs_table *t, gs_reads *r, uint32_t *has_raw, uint32_t *last_loc);
int gs_tag_phase2(gs_table *t, gs_reads *r, uint32_t carry_in);
/* gs_tag_phase1 in two halves: _launch queues the probe kernel and the summary scan and returns at once (the rows of the batch's
 * LAST finished pass stay valid in device memory until its phase 2: a gather of them -- gs_gather_hits -- can run beside the
 * probe kernels of the next pass); _wait blocks until the summary is on the host. */
int gs_tag_phase1_launch(gs_table *t, gs_reads *r);
int gs_tag_phase1_wait(gs_table *t, gs_reads *r, uint32_t *has_raw, uint32_t *last_loc);
/* Copy the emitted hits of the last pass again, to host memory or into caller-provided DEVICE
 * memory (for a collective); n_out as in gs_tag_finish. */
int gs_tag_hits_to_host(gs_table *t, gs_reads *r, gs_hit *out, uint64_t cap, uint64_t *n_out);
/* The same without a copy: the emit kernel writes every row to pinned HOST memory as well as to the device list; *out points
 * at those *n_out rows.  Valid until the next pass over `r` or its destruction. */
int gs_tag_hits_view(gs_table *t, gs_reads *r, const gs_hit **out, uint64_t *n_out);
int gs_tag_hits_to_device(gs_table *t, gs_reads *r, void *dst_device, uint64_t cap, uint64_t *n_out);
/* Several batches in flight against one table: the pass of `r` is queued behind the pass of `prev` (launched before, on the
 * same table) and takes its `prevLoc` from it ON THE DEVICE -- kmerpos_annot3.nim:82-84: the state crosses reads, and
 * therefore batches -- so no host round trip separates them.  Every batch is finished on its own (gs_tag_finish /
 * gs_tag_hits_view), in any order; a batch's pass state lives in its gs_reads handle. */
int gs_tag_launch_after(gs_table *t, gs_reads *r, const gs_reads *prev);

/* ---- streaming engine: `for rec in readfq(reads)` + the main loop (kmerpos_annot3.nim:85-96) as a pipeline of batches ----
 * gs_stream_create   `depth` slots sized once for batches of up to max_words packed words / max_reads reads (max_words = 0:
 *                    no slots, a stream of caller-owned device batches only); carry_in = `prevLoc` before the first batch.
 * gs_stream_submit   a host batch in the gs_reads_upload layout.  Returns as soon as the copies and kernels are queued: the
 *                    upload of this batch overlaps the pass of the one before it, `prevLoc` goes from pass to pass in device
 *                    memory.  `words` in pinned memory (gs_host_alloc) are read by DMA where they lie and must stay untouched
 *                    until the batch comes back from gs_stream_next; pageable words are copied before the call returns.
 *                    GS_E_CAPACITY: every slot is in flight or held -- call gs_stream_next / gs_stream_release first.
 * gs_stream_submit_reads   the same for a batch that already is device-resident (a gs_reads handle the caller keeps).
 * gs_stream_next     the OLDEST batch in flight: waits for ITS pass (later ones keep running), returns its rows in pinned host
 *                    memory (*hits, valid until the next gs_stream_next or gs_stream_release), its read count and counters.
 *                    GS_E_KEYERROR hands out the rows before the offending window and stops the stream (the reference exits
 *                    there: kmerpos_annot3.nim:90).
 * gs_stream_carry    `prevLoc` after the last batch gs_stream_next returned. */
typedef struct gs_stream gs_stream;
int gs_stream_create(gs_table *t, uint64_t max_words, uint32_t max_reads, int depth, uint32_t carry_in, gs_stream **out);
void gs_stream_destroy(gs_stream *s);
int gs_stream_submit(gs_stream *s, const uint64_t *words, uint64_t n_words, const uint64_t *word_off, const uint32_t *len,
                     uint32_t n_reads);
int gs_stream_submit_reads(gs_stream *s, gs_reads *r);
int gs_stream_next(gs_stream *s, const gs_hit **hits, uint64_t *n_hits, uint32_t *n_reads, gs_counters *ctr);
void gs_stream_release(gs_stream *s);
int gs_stream_in_flight(const gs_stream *s);
uint32_t gs_stream_carry(const gs_stream *s);
/* pinned host memory for batches handed to gs_stream_submit (any host memory works; this kind is not copied a second time) */
int gs_host_alloc(uint64_t bytes, void **out);
void gs_host_free(void *p);
void gs_counters_reset(gs_table *t);
/* the table's counters as accumulated over every pass settled since the last reset (windows / raw_hits / emitted / tiles: of
 * the pass settled last) */
void gs_table_counters(const gs_table *t, gs_counters *out);
/* packed words [word_lo, word_lo + n_words) of a resident batch back on the host (the gs_reads_upload layout) */
int gs_reads_copy_words(const gs_reads *r, uint64_t word_lo, uint64_t n_words, uint64_t *dst);


/* The output file: rows `name \t i \t chrom \t start \t group \n` (kmerpos_annot3.nim:93) formatted and written natively.
 * Names of a batch come from the text block a device-parsed batch was made from (names_base + name_off[read], name_len[read]:
 * gs_textindex_name_off / _name_len) or, with name_off = NULL, from a host-parsed `batch`. */
typedef struct gs_rowwriter gs_rowwriter;
int gs_rowwriter_open(const char *path, gs_rowwriter **out);
int gs_rowwriter_rows(gs_rowwriter *w, const gs_hit *hits, uint64_t n, const gs_sunkdb *db, const uint8_t *names_base,
                      const uint64_t *name_off, const uint32_t *name_len, const gs_batch *batch);
int gs_rowwriter_close(gs_rowwriter *w, uint64_t *rows_out);
/* `name \t len` per record (workflow/src/rlen.nim:13-14, rule read_lengths, tagONT.smk:73) for a block of text holding whole records
 * in a regular form (FASTA; four-line FASTQ), appended to the writer's file: a scan of the text, nothing packed or copied.
 * GS_E_IRREGULAR for any text gs_fastx_* (kseq's rules) might read differently: the caller then reads the file with gs_fastx_*. */
int gs_rowwriter_rlen(gs_rowwriter *w, const uint8_t *text, uint64_t n, uint64_t *n_records);
/* The same bytes for the rows of the LAST PASS over `r`, whose hit list is still in device memory: measured, placed and written
 * as text by kernels (csrc/gs_format.hip; the loc record of a row is one 16-byte gather from `lr`), the finished text handed to the
 * file in 32 MiB pieces.  Names: (names_base + name_off[read], name_len[read]) for the n_names reads of the batch, or `batch` when
 * name_off is NULL.  contig_names[n_contigs] = the names behind the loc rows' contig ids (gs_sunkdb_contig_name). */
struct gs_locrows;
int gs_rowwriter_rows_device(gs_rowwriter *w, gs_table *t, gs_reads *r, const struct gs_locrows *lr, uint32_t n_contigs,
                             const char *const *contig_names, const uint8_t *names_base, const uint64_t *name_off, const uint32_t *name_len,
                             uint32_t n_names, const gs_batch *batch);
/* ... and where in this call's text given rows start: byte_cuts[i] = bytes of the rows in front of row row_cuts[i] of the pass
 * (ascending, at most the pass's row count, which gives the call's total).  For --emulate-chunks (tagONT.smk:2-37,112-131 in one
 * run): a batch tagged in chunk-major order is written as one text and every scatter chunk's stretch of it is known. */
int gs_rowwriter_rows_device_cuts(gs_rowwriter *w, gs_table *t, gs_reads *r, const struct gs_locrows *lr, uint32_t n_contigs,
                                  const char *const *contig_names, const uint8_t *names_base, const uint64_t *name_off, const uint32_t *name_len,
                                  uint32_t n_names, const gs_batch *batch, const uint64_t *row_cuts, uint32_t n_cuts, uint64_t *byte_cuts);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU: read-sharded tagging, the SUNK table replicated.  One process per GPU; the data plane is
 * RCCL over xGMI, inside the library.  The reference has no counterpart (its parallelism is Snakemake's per-chunk scatter,
 * workflow/Snakefile:23-24, tagONT.smk:19-37); the exchange exists because `prevLoc` is never reset between reads
 * (kmerpos_annot3.nim:82-84):
 *   every rank   gs_tag_phase1 -> {has a raw hit, loc of the last one}                       (8 bytes per rank: the caller's
 *                control plane all-gathers them; carry-in = last loc of the nearest earlier shard with a raw hit)
 *                gs_tag_phase2(carry-in), gs_tag_finish
 *   all ranks    gs_gather_hits: the per-rank hit lists back to back on the root, read indices shifted to the whole input
 * ---------------------------------------------------------------------------------------- */
#define GS_COMM_ID_BYTES 128
typedef struct gs_comm gs_comm;
/* rank 0 creates the id (ncclGetUniqueId) and hands its GS_COMM_ID_BYTES bytes to the other ranks out of band */
int gs_comm_unique_id(void *id_out);
/* collective over all ranks (ncclCommInitRank); `device` is this rank's GPU */
int gs_comm_init(int device, int nranks, int rank, const void *unique_id, gs_comm **out);
void gs_comm_destroy(gs_comm *c);
/* The root's BUILT table, device array by device array, to every other rank (ncclBroadcast): nobody else parses or builds.
 * The root passes its table and gets it back in *out; the others pass NULL and own the table they receive. */
int gs_table_broadcast(gs_comm *c, int root, gs_table *mine, gs_table **out);
/* What a receiving rank of gs_table_broadcast does -- the root's scalars, every array allocated from them (all of them before
 * any transfer: a rank that cannot allocate fails ALL ranks instead of leaving them in a collective), a handle initialised
 * without a build -- with device-to-device copies on ONE GPU: a second, independent handle of the same table. */
int gs_table_clone(const gs_table *src, gs_table **out);
/* ncclAllGather of n (1 or 2) 64-bit words per rank on the library's stream: the phase-1 summaries {has a raw hit, loc of the
 * last one} of a read-sharded pass without a second transport.  out[rank * n + i], host. */
int gs_comm_allgather_u64(gs_comm *c, const uint64_t *mine, uint32_t n, uint64_t *out);
/* The two exchanges of a read-sharded SECOND half (every rank filters and pair-tests its own rows; the hit lists never leave
 * their GPU): ncclAllReduce (sum, in place) of n uint32 counters in device memory -- the reads per SUNK of the bad-SUNK
 * histogram; gs_comm_reduce_u32 is the same as a gs_reduce_u32_fn with ctx = the communicator -- and a variable-size gather of
 * host bytes to the root (the ranks' TSV rows): sizes and a status word by ncclAllGather first (a failed rank fails all of them
 * before anything is sent), then grouped ncclSend / ncclRecv of exactly each rank's bytes.  sizes[nranks] on every rank; *out
 * (root only; gs_host_free) = the ranks' bytes back to back in rank order. */
int gs_comm_allreduce_u32(gs_comm *c, void *d_buf, uint64_t n);
int gs_comm_reduce_u32(void *ctx, void *d_buf, uint64_t n);
int gs_comm_gatherv_host(gs_comm *c, int root, const void *mine, uint64_t nbytes, int status, void **out, uint64_t *sizes);
/* After a finished pass on every rank: all ranks' emitted hits concatenated in the ROOT'S DEVICE MEMORY, in rank order, `read`
 * shifted by the rank's read_offset.  Counts and offsets travel by ncclAllGather, the rows by grouped ncclSend / ncclRecv of
 * exactly their size.  A rank whose pass failed (GS_E_KEYERROR: kmerpos_annot3.nim:90 is data-dependent, one shard meets it)
 * still takes part in the exchange of the counts, which carries its status: EVERY rank then returns that code and no row is
 * sent -- an N-GPU run exits 1 like the reference instead of hanging.  Root: *n_out = total rows; `out` (host, `cap` rows) may be NULL: the rows then stay on the device
 * (gs_gather_device) and cross PCIe only if gs_gather_view is called -- every rank's own rows already are on its own host
 * after gs_tag_finish.  Other ranks pass out = NULL. */
int gs_gather_hits(gs_comm *c, int root, gs_table *t, gs_reads *r, uint32_t read_offset, gs_hit *out, uint64_t cap, uint64_t *n_out);
/* Root, after gs_gather_hits: the gathered rows in device memory, valid until the next gather. */
int gs_gather_device(gs_comm *c, const void **d_out, uint64_t *n_out);
/* Root, after gs_gather_hits: the gathered rows in pinned host memory of the library (copied there on the first call after a
 * gather), valid until the next gather. */
int gs_gather_view(gs_comm *c, const gs_hit **out, uint64_t *n_out);


/* ------------------------------------------------------------------------------------------
 * Prebuilt sidecar of a (k-mer list, location table) pair.  `kmerpos_annot3 <reads> <db> <loc> <out>` parses both text files on
 * every invocation (kmerpos_annot3.nim:20-72) and rule SUNK_annot runs once per scatter chunk (tagONT.smk:19-37): at genome
 * scale that is 1.5e9 rows of text, 10 x 2 times per sample.  gs_sidecar_write keeps the parsed location rows and the BUILT
 * device table in one binary file; gs_sidecar_open hands them back -- only while the file's fingerprints of <db> and <loc>
 * (size, modification time, hashes of the first and last MiB) still hold; GS_E_IO = none / stale / other version: parse and
 * build as before.  The gs_sunkdb it returns holds what a hit is formatted from (k, contig names, row_contig / row_start /
 * row_group); its key arrays are empty (they live in the table).
 * ---------------------------------------------------------------------------------------- */
int gs_sidecar_write(const char *path, const gs_sunkdb *db, const gs_table *t, const char *db_path, const char *loc_path);
int gs_sidecar_open(const char *path, const char *db_path, const char *loc_path, int device, gs_sunkdb **db_out, gs_table **t_out);


/* ------------------------------------------------------------------------------------------
 * Diagonal filter (workflow/src/diag_filter_v3.nim + diag_filter_step2.nim).
 * ---------------------------------------------------------------------------------------- */

/* A `.sunkpos` file (rows `rname \t pos \t contig \t start \t group`, plain or gzip) parsed into
 * arrays.  `seg` numbers the runs of consecutive rows with the same rname: the reference flushes
 * whenever the name changes (diag_filter_v3.nim:64), so a name that comes back later is a new read. */
typedef struct gs_sunkpos gs_sunkpos;
int gs_sunkpos_load(const char *path, gs_sunkpos **out);
/* `diag_filter_step2 <sunkpos> <diag>` as the reference program does it on files (diag_filter_step2.nim:16-64; rule
 * diag_filter_final, tagONT.smk:110): the `.sunkpos` rows whose contig is the best contig the diag file records for their
 * read, written to out_path (NULL: stdout).  Both inputs plain or gzip; no GPU involved (a text join; inside one process
 * gs_diag_filter's keep[] is the same mask).  *rows_out = rows written. */
int gs_diag_filter_step2_files(const char *sunkpos_path, const char *diag_path, const char *out_path, uint64_t *rows_out);
void gs_sunkpos_free(gs_sunkpos *s);
uint64_t gs_sunkpos_n_rows(const gs_sunkpos *s);
uint32_t gs_sunkpos_n_segs(const gs_sunkpos *s);
const uint32_t *gs_sunkpos_row_seg(const gs_sunkpos *s);
const int64_t *gs_sunkpos_row_pos(const gs_sunkpos *s);
const uint32_t *gs_sunkpos_row_contig(const gs_sunkpos *s);
const int64_t *gs_sunkpos_row_start(const gs_sunkpos *s);
const int64_t *gs_sunkpos_row_group(const gs_sunkpos *s);
const char *gs_sunkpos_seg_name(const gs_sunkpos *s, uint32_t seg);
uint32_t gs_sunkpos_n_contigs(const gs_sunkpos *s);
const char *gs_sunkpos_contig_name(const gs_sunkpos *s, uint32_t contig);
/* the text of row i (without newline), for step 2's verbatim copy-through */
const char *gs_sunkpos_row_text(const gs_sunkpos *s, uint64_t row, uint32_t *len);

/* What the row-set functions below need from a `.sunkpos` file beside its columns (all host side, native):
 *   the distinct read names in first-seen order -- name id of every row, rank of every name in sorted order (what
 *   `groupby("rname")` iterates in, process-by-contig_lowmem_AR.py:135), the name's text;
 *   gs_sunkpos_names_long  name_long[n_names] <- 1 where the `.rlen` file has a row of that name with len >= minlen (:106-108)
 *   gs_sunkpos_rows_bad    row_bad[n_rows] <- 1 where the row's `chrom:ID` is a line of the bad-SUNK list (:66-72)
 *   gs_sunkpos_row_keys    dense ids of the rows' distinct (chrom, ID) pairs (first-seen order) -> number of keys */
uint32_t gs_sunkpos_n_names(gs_sunkpos *s);
const uint32_t *gs_sunkpos_row_name(gs_sunkpos *s);
const uint32_t *gs_sunkpos_name_rank(gs_sunkpos *s);
const char *gs_sunkpos_name(gs_sunkpos *s, uint32_t name);
int gs_sunkpos_names_long(gs_sunkpos *s, const char *rlen_path, int64_t minlen, uint8_t *name_long);
int gs_sunkpos_rows_bad(gs_sunkpos *s, const char *bad_path, uint8_t *row_bad);
uint32_t gs_sunkpos_row_keys(gs_sunkpos *s, uint32_t *row_key, const uint32_t **key_contig, const int64_t **key_id);

/* Per segment (read): the contig with the most distinct SUNK groups within 2500 bp of the median
 * diagonal, trying `pos+start` ("-") then `start-pos` ("+") per contig; ties go to the contig
 * with fewer hits, then to the first contig in the slot order of Nim's Table[string, ...]
 * (diag_filter_v3.nim:76-138).  Only contigs with in_fai[contig] != 0 count (:147-150).
 *   best_contig[seg] = contig id or GS_NONE when the read prints nothing (maxgood <= 1, :141)
 *   best_good[seg]   = maxgood,  best_dir[seg] = '-' or '+'
 *   keep[row]        = 1 when diag_filter_step2 keeps the row (contig == best contig of its read,
 *                      diag_filter_step2.nim:45-64)
 * The per-(read, contig) statistics and the selection run on the GPU; the host walks the rows once
 * to lay out segments and to replay the Table's slot order (a serial hash-table emulation). */
int gs_diag_filter(int device, const gs_sunkpos *s, const uint8_t *in_fai, uint32_t *best_contig,
                   uint32_t *best_good, uint8_t *best_dir, uint8_t *keep);


/* ------------------------------------------------------------------------------------------
 * Inter-SUNK distance test (workflow/scripts/process-by-contig_lowmem_AR.py:135-200).
 * ---------------------------------------------------------------------------------------- */

/* Per read: all pairs of its SUNK hits whose distance on the read is within (0.9, 1.1) of their
 * distance on the contig (fp64, as scipy.pdist + numpy compute it, :140-147), restricted to the
 * majority orientation (:150-153), cleaned of IDs seen at several read positions (:160-181), then
 * the largest connected component of the ID graph; its IDs in vertex insertion order (:189-198).
 *   read_off[n_reads+1] : row ranges; rows of a read sorted by `start` (stable) and de-duplicated,
 *                         i.e. `sub` of the script (lines 50-108 are file-level filtering, host side)
 *   out_n[r]            : number of IDs kept for read r (0 = the read prints nothing)
 *   out_ids             : n_rows slots; read r's IDs start at out_ids[read_off[r]] */
int gs_pair_component(int device, uint32_t n_reads, const uint64_t *read_off, const int64_t *pos,
                      const int64_t *start, const int64_t *id, uint32_t *out_n, int64_t *out_ids);


/* Bad-SUNK histogram (workflow/scripts/badsunks_AR.py:20-51).
 *   row_key[n_rows]  dense id of each .sunkpos row's `chrom:ID`
 *   key_sel[n_keys]  1 for keys on this haplotype's contigs ("correct_hap", :31-34)
 *   key_count[n_keys] <- reads per key (`value_counts`, :23-24)
 *   hist[hist_len]    <- number of selected keys per count (last bin = that count or more); its arg-max is
 *                        the script's "mean coverage" = mode of the counts (:43) */
int gs_key_histogram(int device, uint64_t n_rows, const uint32_t *row_key, uint32_t n_keys, const uint8_t *key_sel,
                     uint32_t *key_count, uint32_t *hist, uint32_t hist_len);


/* ------------------------------------------------------------------------------------------
 * SUNK definition on the GPU: the table's input straight from the
 * assemblies, without the text round trip through jellyfish.db / kmer.loc.
 * ---------------------------------------------------------------------------------------- */

/* Canonical k-mers that occur exactly once over all contigs (`jellyfish count -C -U 1`,
 * workflow/rules/defineSUNKs.smk:39; windows containing a non-ACGT byte are not k-mers), their
 * positions (`mrsfast -e 0`, :101) and groups = merged overlapping or book-ended [start, start+k)
 * intervals, id = start of the merged interval (`bedtools merge`, :124-126).
 *   seq/off : ASCII bases of all contigs (hap1 then hap2) back to back, off[n_contigs+1]
 * Result rows are in `kmer.loc` order, (contig, start) ascending. */
typedef struct gs_sunkset gs_sunkset;
int gs_sunks_define(int device, int k, uint32_t n_contigs, const uint8_t *seq, const uint64_t *off, gs_sunkset **out);
void gs_sunkset_free(gs_sunkset *s);
uint64_t gs_sunkset_n(const gs_sunkset *s);
const uint64_t *gs_sunkset_code(const gs_sunkset *s);
const uint32_t *gs_sunkset_contig(const gs_sunkset *s);
const int64_t *gs_sunkset_start(const gs_sunkset *s);
const int64_t *gs_sunkset_group(const gs_sunkset *s);


/* The same with the assembly already in DEVICE memory and the result kept there: feed it to gs_table_create_device
 * (keys = code, loc = loc, payload NULL: a hit's payload is then the kmer.loc row).  Arrays (device pointers, valid until
 * gs_sunkset_dev_free), all in kmer.loc order:  code u64 | pos u64 = contig << 32 | start | group u32 = start of the merged
 * interval | loc u32 = dense id of the (contig, group) pair. */
typedef struct gs_sunkset_dev gs_sunkset_dev;
int gs_sunks_define_device(int device, int k, uint32_t n_contigs, const void *d_seq, const uint64_t *off, gs_sunkset_dev **out);
/* <db> (one canonical k-mer per line, ascending code) and <loc> (`contig \t start \t k-mer \t group` rows in (contig, start)
 * order) of a device-resident SUNK set: the two files `kmerpos_annot3` reads (defineSUNKs.smk:59,126), written natively */
int gs_sunkset_dev_write_files(const gs_sunkset_dev *s, const char *const *contig_names, uint32_t n_names, const char *db_path,
                               const char *loc_path);
void gs_sunkset_dev_free(gs_sunkset_dev *s);
uint64_t gs_sunkset_dev_n(const gs_sunkset_dev *s);
uint64_t gs_sunkset_dev_n_groups(const gs_sunkset_dev *s);
const void *gs_sunkset_dev_code(const gs_sunkset_dev *s);
const void *gs_sunkset_dev_pos(const gs_sunkset_dev *s);
const void *gs_sunkset_dev_group(const gs_sunkset_dev *s);
const void *gs_sunkset_dev_loc(const gs_sunkset_dev *s);
/* plain device-to-host copy for callers that hold such device pointers but no HIP binding of their own */
int gs_copy_to_host(int device, void *dst, const void *d_src, uint64_t bytes);


/* Contig-wide union of the reads' validated SUNK sets (process-by-contig_lowmem_AR.py:219-260; SURVEY.md
 * section 8f row N3).  Rows = the `ID \t rname` output rows; row_read numbers the reads (rows of a read are
 * consecutive), row_vertex is the dense index of the row's ID, vertex_id the ID value of each vertex.
 * Per vertex v that is the root of a component: comp_size[v] = SUNKs in it (0 for non-roots),
 * comp_min / comp_max = its smallest / largest ID.  Components with comp_size > 2 are the validated
 * intervals of the BED (:247-251). */
int gs_union_regions(int device, uint64_t n_rows, const uint32_t *row_read, const uint32_t *row_vertex,
                     uint32_t n_vertices, const int64_t *vertex_id, uint32_t *comp_size, int64_t *comp_min,
                     int64_t *comp_max);


/* ------------------------------------------------------------------------------------------
 * The device-resident second half of the engine: hit list (HBM) -> rows -> diagonal filter -> bad-SUNK histogram -> the
 * file-level filtering and the pair stage of process-by-contig -> contig-wide union, without leaving the device.  The file
 * drop-ins above (diag_filter_v3, badsunks, process-by-contig) run the same functions over rows parsed from text.
 * ---------------------------------------------------------------------------------------- */

/* The rows of the last pass over `r` where they lie in DEVICE memory (gs_hit[*n_out], stream order); valid until the next
 * pass over `r`.  What kmerpos_annot3.nim:93 prints, before any text exists. */
int gs_tag_hits_device(gs_table *t, gs_reads *r, const void **d_hits, uint64_t *n_out);

/* kmer.loc rows in device memory: what turns a hit's payload (the loc row) into `chrom, start, ID` and the dense id of
 * `chrom:ID` (kmerpos_annot3.nim:44-69 keeps them in `coords`).  From the host arrays of a gs_sunkdb (row_contig / row_start /
 * row_group / row_loc), or straight from a device-resident SUNK set.  Either way the handle owns ONE packed 16-byte record per row
 * {start, contig, ID, key} (a row of a hit list is one random 16-byte load); it does not refer to the set afterwards. */
typedef struct gs_locrows gs_locrows;
int gs_locrows_create(int device, uint64_t n_rows, const uint32_t *row_contig, const int64_t *row_start, const int64_t *row_group,
                      const uint32_t *row_loc, gs_locrows **out);
int gs_locrows_from_sunkset_dev(int device, const gs_sunkset_dev *s, gs_locrows **out);
void gs_locrows_free(gs_locrows *lr);
uint64_t gs_locrows_n_rows(const gs_locrows *lr);
uint32_t gs_locrows_n_locs(const gs_locrows *lr);
/* contig of every dense (chrom, ID) key: key_contig[gs_locrows_n_locs] (GS_NONE where no row carries the key) */
int gs_locrows_key_contig(const gs_locrows *lr, uint32_t *key_contig);

/* `.sunkpos` rows (`rname \t pos \t chrom \t start \t ID`, kmerpos_annot3.nim:93) as arrays in device memory, file order:
 * name = id of the read name (equal names = equal ids), pos, contig id, start, ID, key = dense id of `chrom:ID`
 * (all uint32: the Python stage reads them as such, process-by-contig_lowmem_AR.py:60).
 *   gs_rowset_append_pass  the rows of the last pass over `r`, device to device (name id = name_base + gs_hit.read)
 *   gs_rowset_append_host  rows parsed from a file (key may be NULL: then no function that needs keys can run) */
typedef struct gs_rowset gs_rowset;
int gs_rowset_create(int device, uint64_t cap_hint, gs_rowset **out);
void gs_rowset_free(gs_rowset *rs);
uint64_t gs_rowset_n(const gs_rowset *rs);
int gs_rowset_append_pass(gs_rowset *rs, gs_table *t, gs_reads *r, uint32_t name_base, const gs_locrows *lr);
int gs_rowset_append_host(gs_rowset *rs, uint64_t n, const uint32_t *name, const uint32_t *pos, const uint32_t *contig,
                          const uint32_t *start, const uint32_t *id, const uint32_t *key);
/* any destination may be NULL */
int gs_rowset_download(const gs_rowset *rs, uint32_t *name, uint32_t *pos, uint32_t *contig, uint32_t *start, uint32_t *id, uint32_t *key);
/* segments = runs of consecutive rows with one name: the reads of diag_filter_v3.nim:64 */
int gs_rowset_segments(gs_rowset *rs, uint32_t *n_segs);
/* diag_filter_v3.nim:63-229 over the row set, everything on the device -- rows sorted by (read, contig), the slot order of Nim's
 * Table[string, ...] (capacity carried from read to read as a running maximum; a read that makes the table grow is replayed
 * exactly), statistics per (read, contig), the selection per read.  contig_names[n_contigs] (the Table hashes the names),
 * in_fai[n_contigs].  Host outputs, n_segs entries each, any may be NULL: seg_name (name id), best_contig (GS_NONE: the read
 * prints nothing), best_good, best_dir ('-' / '+').  Every row gets its verdict (diag_filter_step2.nim:45-64). */
int gs_rowset_diag_filter(gs_rowset *rs, uint32_t n_contigs, const char *const *contig_names, const uint8_t *in_fai, uint32_t *seg_name,
                          uint32_t *best_contig, uint32_t *best_good, uint8_t *best_dir, uint64_t *n_kept);
/* The same for ONE shard of a read-sharded run: the Table's capacity is the only state diag_filter_v3
 * carries from read to read (`clear()` keeps it, diag_filter_v3.nim:143), so a shard needs the capacity the reads BEFORE it
 * leave behind: cap_in (a power of two >= 64; 64 = a fresh Table), and reports the one it leaves: *cap_out (may be NULL).
 * Ranks run it with cap_in = 64, exchange cap_out, and the few whose predecessors grew the Table run it again. */
int gs_rowset_diag_filter_from(gs_rowset *rs, uint32_t n_contigs, const char *const *contig_names, const uint8_t *in_fai, uint32_t cap_in,
                               uint32_t *seg_name, uint32_t *best_contig, uint32_t *best_good, uint8_t *best_dir, uint64_t *n_kept,
                               uint32_t *cap_out);
int gs_rowset_keep_flags(const gs_rowset *rs, uint8_t *keep);
/* diag_filter_step2: drop the rows the filter did not keep (in place, order preserved) */
int gs_rowset_keep_filtered(gs_rowset *rs);
/* badsunks_AR.py:20-103 over the two haplotypes' filtered row sets.  key_hap[n_keys]: bit 0 = the key's contig is in hap1's
 * fai, bit 1 = in hap2's.  bad_host[n_keys] (may be NULL) and *d_bad (may be NULL; device flags, gs_device_free) <- the union
 * of the four bad sets; meancov[h] = the script's "mean coverage" (mode of the reads per SUNK group on the haplotype's own contigs). */
int gs_rowset_bad_keys(const gs_rowset *hap1, const gs_rowset *hap2, uint32_t n_keys, const uint8_t *key_hap, uint8_t *bad_host,
                       uint32_t meancov[2], void **d_bad);
/* The same when the reads are sharded over ranks: the reads-per-SUNK counters (badsunks_AR.py:33-36 counts over ALL reads) are
 * summed over the ranks before the histogram -- `reduce(ctx, d_counts, n_keys)` is called once per haplotype with the uint32
 * counters in device memory and returns with the sums in place on every rank (gs_comm_reduce_u32 with ctx = the communicator:
 * one ncclAllReduce; NULL = one rank). */
typedef int (*gs_reduce_u32_fn)(void *ctx, void *d_counts, uint64_t n);
int gs_rowset_bad_keys_reduced(const gs_rowset *hap1, const gs_rowset *hap2, uint32_t n_keys, const uint8_t *key_hap, gs_reduce_u32_fn reduce,
                               void *reduce_ctx, uint8_t *bad_host, uint32_t meancov[2], void **d_bad);
int gs_device_free(int device, void *p);
/* The second half keeps its device temporaries in a pool between calls (csrc/gs_devpool.h): hand the idle ones back to the driver. */
int gs_device_pool_trim(int device);
/* (tests) makes a kernel launch fail on this thread: the error stays in HIP's last-error slot, as a failed launch of the library's
 * own would; the gs_comm_* entry points look at the slot BEFORE their collectives and fail on it. */
int gs_selftest_inject_launch_error(int device);
int gs_device_copy_to_host(int device, void *host, const void *d_src, uint64_t nbytes);
int gs_device_copy_from_host(int device, void *d_dst, const void *host, uint64_t nbytes);
/* process-by-contig_lowmem_AR.py:50-260 over a row set (rows of several contigs are handled contig by contig, like the
 * workflow's per-contig files): bad SUNKs dropped (d_key_bad: device flags per key, and / or row_bad: host flags per row),
 * reads with >= 2 distinct IDs, sort by (name, start), exact duplicates, reads >= 10 kb (name_long[n_names]), then per read the
 * pair test / orientation / multi-position clean-up / largest component (gs_pair_component's kernels), the `ID \t rname` rows in
 * sorted-name order (name_rank[n_names]) and the contig-wide union (gs_union_regions' kernels) -> validated intervals. */
typedef struct gs_pairstage gs_pairstage;
int gs_rowset_pair_stage(const gs_rowset *rs, const void *d_key_bad, const uint8_t *row_bad, uint32_t n_names, const uint32_t *name_rank,
                         const uint8_t *name_long, gs_pairstage **out);
/* Root of a read-sharded run: the TSV rows of ALL ranks (sorted by contig, then by the reads' name order; a read's rows adjacent,
 * in the order its rank produced them) -> the result a one-GPU run over all reads gives: the rows as handed in, the validated
 * intervals from the contig-wide union over all of them (:219-260 is commutative over reads).  contigs_in: the union of the
 * ranks' gs_pairstage_contigs_in. */
int gs_pairstage_from_rows(int device, uint64_t n_rows, const uint32_t *row_contig, const uint32_t *row_name, const int64_t *row_id,
                           uint32_t n_contigs_in, const uint32_t *contigs_in, gs_pairstage **out);
void gs_pairstage_free(gs_pairstage *ps);
uint64_t gs_pairstage_n_rows(const gs_pairstage *ps);            /* TSV rows, contig by contig (ascending id), reads in name order */
const uint32_t *gs_pairstage_row_contig(const gs_pairstage *ps);
const uint32_t *gs_pairstage_row_name(const gs_pairstage *ps);
const int64_t *gs_pairstage_row_id(const gs_pairstage *ps);
uint32_t gs_pairstage_n_regions(const gs_pairstage *ps);         /* BED rows, by (contig, start) */
const uint32_t *gs_pairstage_region_contig(const gs_pairstage *ps);
const int64_t *gs_pairstage_region_start(const gs_pairstage *ps);
const int64_t *gs_pairstage_region_end(const gs_pairstage *ps);
uint32_t gs_pairstage_n_contigs_in(const gs_pairstage *ps);      /* distinct contigs of the input rows (a contig without output
                                                                    still gets its one-line file, :92-94,202-204) */
const uint32_t *gs_pairstage_contigs_in(const gs_pairstage *ps);
/* rows in | not bad | in reads with >= 2 IDs | after de-duplication | in reads >= 10 kb | read groups | TSV rows | 0 */
const uint64_t *gs_pairstage_counters(const gs_pairstage *ps);

#ifdef __cplusplus
}
#endif
#endif /* GAVISUNK_H */
