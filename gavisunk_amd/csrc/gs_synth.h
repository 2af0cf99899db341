/* libgssynth.so -- synthetic genome-scale inputs generated on the GPU: BENCH / TEST SUPPORT only.
 * Not part of the drop-in boundary (include/gavisunk.h); the product path never loads it.  See gs_synth.hip.
 * All functions return 0 or a negative status; gs_synth_last_error() explains.  Device buffers returned through
 * `void **` are hipMalloc'ed on `device` and are released with gs_synth_free. */
#ifndef GS_SYNTH_H
#define GS_SYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
const char *gs_synth_last_error(void);
void gs_synth_free(int device, void *d);
/* ASCII assembly of n_bases: uniform ACGT, slots of family_len bases are repeat-family copies with probability repeat_fraction */
int gs_synth_genome(int device, uint64_t seed, uint64_t n_bases, double repeat_fraction, uint32_t n_families, uint32_t family_len,
                    uint32_t break_period, void **d_ascii_out);
/* diploid assembly (BASELINE.json configs[3]): hap 1 then hap 2, n_bases_per_hap each, both descended from the assembly
 * gs_synth_genome makes from the same arguments; the haplotypes differ at a fraction `snp` of the sites, and a fraction `spec`
 * of each haplotype's slots of spec_len bases is sequence found nowhere else */
int gs_synth_diploid(int device, uint64_t seed, uint64_t n_bases_per_hap, double repeat_fraction, uint32_t n_families,
                     uint32_t family_len, double snp, double spec, uint32_t spec_len, void **d_ascii_out);
/* reads cut from [src_start, src_start + src_len) of the assembly (reverse-complemented where rev), with per-base
 * substitution / deletion / insertion; off_out[n_reads + 1] (host) receives the offsets of the reads in *d_ascii_out */
int gs_synth_reads(int device, uint64_t seed, const void *d_genome_ascii, uint32_t n_reads, const uint64_t *src_start,
                   const uint32_t *src_len, const uint8_t *rev, double sub, double dele, double ins, void **d_ascii_out,
                   uint64_t *off_out);
/* indices i of d_codes[0..n) (u64, device) whose value occurs in sorted_queries (host); unordered */
int gs_synth_select_rows(int device, const void *d_codes, uint64_t n, const uint64_t *sorted_queries, uint64_t nq, uint32_t *rows_out,
                         uint64_t cap, uint64_t *n_rows);
/* out[i] = d_src[rows[i]] for 4- or 8-byte elements (device source, host rows and result) */
int gs_synth_gather(int device, const void *d_src, int elem_bytes, const uint32_t *rows, uint64_t n, void *out);
#ifdef __cplusplus
}
#endif
#endif
