// csrc/gs_ingest.hip -- the record parser that reads UNTRUSTED text on the device: its host code and its kernels, the product's source
// as it is -- compiled for the CPU behind tools/hip_cpu and run under AddressSanitizer + UBSan by tools/ingest_cpu_asan.sh.  TEST
// INFRASTRUCTURE (the GPU pool refuses sanitizer runs).
//   ingest_cpu_main <text file> ...   every file goes through gs_reads_from_text; a text it takes must give the names, lengths and
//   bases the host record reader with kseq's rules (gs_fastx_*, csrc/gs_host.cpp) gives for the same file; a text it refuses
//   (GS_E_IRREGULAR) is the host reader's to read: nothing to compare.
#include <string>

#include "../gavisunk_amd/csrc/gs_ingest.hip"

extern "C" int gs_device_count(void) { return 1; }  // (gs_device.hip's)

// what the parser hands to the packer of gs_device.hip: the gathered bases and the reads' offsets in them
static std::string g_seq;
static std::vector<uint64_t> g_off;
int gs::reads_from_parsed_text(int, const void *d_seq, const uint64_t *off, uint32_t n_reads, int, uint64_t *, gs_reads **out) {
    g_off.assign(off, off + n_reads + 1);
    g_seq.assign(static_cast<const char *>(d_seq), (size_t)off[n_reads]);
    *out = reinterpret_cast<gs_reads *>(&g_seq);  // (never looked into)
    return GS_OK;
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    int bad = 0, taken = 0, refused = 0;
    for (int a = 1; a < argc; a++) {
        std::vector<uint8_t> text;
        {
            FILE *f = fopen(argv[a], "rb");
            if (!f) {
                printf("MISSING %s\n", argv[a]);
                bad++;
                continue;
            }
            uint8_t buf[1 << 16];
            size_t n;
            while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.insert(text.end(), buf, buf + n);
            fclose(f);
        }
        // (a copy of exactly the text's size: the staging code must not read past what the caller gave it)
        uint8_t *exact = static_cast<uint8_t *>(malloc(text.size() ? text.size() : 1));
        memcpy(exact, text.data(), text.size());
        gs_reads *r = nullptr;
        gs_textindex *ix = nullptr;
        const int rc = gs_reads_from_text(0, exact, text.size(), 20, &r, &ix);
        if (rc == GS_E_IRREGULAR) {
            refused++;
            printf("ok %s: refused (%s)\n", argv[a], gs_last_error());
            free(exact);
            continue;
        }
        if (rc != GS_OK) {
            printf("MISMATCH %s: rc %d: %s\n", argv[a], rc, gs_last_error());
            bad++;
            free(exact);
            continue;
        }
        taken++;
        // the host reader on the same file
        gs_fastx *fx = nullptr;
        std::string why;
        if (gs_fastx_open(argv[a], &fx) != GS_OK) why = "the host reader cannot open the file";
        uint32_t at = 0;
        const uint32_t n = gs_textindex_n_reads(ix);
        while (why.empty()) {
            gs_batch *b = nullptr;
            const int rb = gs_fastx_next_batch(fx, (uint64_t)1 << 26, 1u << 20, &b);
            if (rb != GS_OK) {
                why = std::string("the host reader fails: ") + gs_last_error();
                break;
            }
            if (!b) break;
            const uint32_t nb = gs_batch_n_reads(b);
            if (!nb) {
                gs_batch_free(b);
                break;
            }
            if (at + nb > n) why = "the host reader finds more reads";
            // the same reads, packed by the same host code from the parser's bases
            gs_batch *mine = nullptr;
            if (why.empty()) {
                std::vector<uint64_t> off(nb + 1);
                for (uint32_t i = 0; i <= nb; i++) off[i] = g_off[at + i] - g_off[at];
                if (gs_batch_from_ascii(reinterpret_cast<const uint8_t *>(g_seq.data()) + g_off[at], off.data(), nb, &mine) != GS_OK) why = "gs_batch_from_ascii fails";
            }
            for (uint32_t i = 0; i < nb && why.empty(); i++) {
                const std::string name(reinterpret_cast<const char *>(exact) + gs_textindex_name_off(ix)[at + i], gs_textindex_name_len(ix)[at + i]);
                if (name != gs_batch_name(b, i)) why = "read " + std::to_string(at + i) + ": name '" + name + "' against '" + gs_batch_name(b, i) + "'";
                else if (gs_textindex_len(ix)[at + i] != gs_batch_len(b)[i]) why = "read " + std::to_string(at + i) + ": length";
            }
            if (why.empty() && (gs_batch_n_words(mine) != gs_batch_n_words(b) || memcmp(gs_batch_words(mine), gs_batch_words(b), gs_batch_n_words(b) * 8) != 0)) why = "bases differ";
            if (mine) gs_batch_free(mine);
            gs_batch_free(b);
            at += nb;
        }
        if (why.empty() && at != n) why = "the host reader finds " + std::to_string(at) + " reads, the parser " + std::to_string(n);
        if (fx) gs_fastx_close(fx);
        gs_textindex_free(ix);
        free(exact);
        printf("%s %s: %u reads%s%s\n", why.empty() ? "ok" : "MISMATCH", argv[a], n, why.empty() ? "" : ": ", why.c_str());
        if (!why.empty()) bad++;
    }
    printf("ingest_cpu_main: %d taken, %d refused, %d bad\n", taken, refused, bad);
    return bad ? 1 : 0;
}
