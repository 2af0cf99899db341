// csrc/gs_bgzf.hip -- host code AND kernels, the product's source as it is -- compiled for the CPU behind tools/hip_cpu (a stand-in
// for the HIP runtime: device memory = malloc'ed blocks of the exact sizes, a block's threads = real threads) and run under
// AddressSanitizer + UBSan by tools/bgzf_cpu_asan.sh.  TEST INFRASTRUCTURE: what the GPU pool cannot do (it refuses sanitizer runs) for
// the one kernel that decodes UNTRUSTED bytes.
//   bgzf_cpu_main <file> ...   every file is read through gs_bgzf_open / gs_bgzf_next with two block sizes and compared with zlib's
//   gzread of the same file: the same bytes where zlib accepts the file, an error (or zlib's bytes) where it does not.
#include <zlib.h>

#define GS_BGZF_NL_BLOCKS 4
#define GS_BGZF_OCC
#include "../gavisunk_amd/csrc/gs_bgzf.hip"

extern "C" int gs_device_count(void) { return 1; }  // (gs_device.hip's)

// the staged-text buffers of gs_ingest.hip: here exactly as large as asked for, so that one byte too many is seen
uint8_t *gs::stage_buffer(int, size_t want, size_t *cap) {
    *cap = want;
    return static_cast<uint8_t *>(malloc(want ? want : 1));
}
void gs::stage_buffer_release(int, uint8_t *p, size_t) { free(p); }
gs_textstage *gs::stage_adopt(int device, uint8_t *d_text, uint32_t n, size_t cap, uint8_t first) {
    auto *s = new gs_textstage();
    s->device = device;
    s->d_text = d_text;
    s->n = n;
    s->cap = cap;
    s->first = first;
    return s;
}

static bool zlib_text(const char *path, std::string *out) {
    gzFile f = gzopen(path, "rb");
    if (!f) return false;
    std::vector<char> buf(1 << 20);
    bool ok = true;
    for (;;) {
        const int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n < 0) {
            ok = false;
            break;
        }
        if (!n) break;
        out->append(buf.data(), (size_t)n);
    }
    if (gzclose(f) != Z_OK) ok = false;
    return ok;
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    int bad = 0;
    for (int a = 1; a < argc; a++) {
        std::string want_text;
        const bool zok = zlib_text(argv[a], &want_text);
        for (const uint64_t want : {(uint64_t)300000, (uint64_t)64 << 20}) {
            gs_bgzf *z = nullptr;
            int rc = gs_bgzf_open(argv[a], 0, &z);
            if (rc == GS_E_IRREGULAR) {
                printf("skip %s: %s\n", argv[a], gs_last_error());
                break;
            }
            std::string got;
            uint64_t blocks = 0, ask = want;
            while (rc == GS_OK) {
                gs_textstage *st = nullptr;
                rc = gs_bgzf_next(z, ask, &st);
                if (rc != GS_OK) break;
                if (!st) {
                    if (gs_bgzf_at_end(z)) break;
                    ask *= 2;  // (not one whole record fit)
                    continue;
                }
                got.append(reinterpret_cast<const char *>(st->d_text), st->n);
                blocks++;
                free(st->d_text);
                delete st;
            }
            if (z) gs_bgzf_close(z);
            const bool same = rc == GS_OK && got == want_text;
            // a damaged file: an error, or -- where the damage is outside what either reader checks -- zlib's bytes
            // (GS_E_IRREGULAR: the file leaves bgzip's container -- a BSIZE that lies --, the caller reads it with the host reader)
            const bool fine = rc == GS_E_IRREGULAR || (zok ? same : (rc != GS_OK || same));
            printf("%s %s want %llu: zlib %s (%zu bytes), here rc %d, %zu bytes in %llu blocks%s%s\n", fine ? "ok" : "MISMATCH", argv[a], (unsigned long long)want,
                   zok ? "ok" : "error", want_text.size(), rc, got.size(), (unsigned long long)blocks, rc != GS_OK ? ": " : "", rc != GS_OK ? gs_last_error() : "");
            if (!fine) bad++;
        }
    }
    return bad ? 1 : 0;
}
