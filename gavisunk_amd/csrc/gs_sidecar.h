// Prebuilt sidecar of a (k-mer list, location table) pair: what workflow/src/kmerpos_annot3.nim:20-72 parses on EVERY invocation
// -- rule SUNK_annot runs once per scatter chunk (workflow/rules/tagONT.smk:19-37), i.e. 10 x 2 times per sample over the same
// two text files, 1.5e9 rows each at genome scale -- kept in binary next to them: the location rows a hit is formatted from
// (contig, start, group; contig names) and the BUILT device table (the same scalars + array list a table broadcast moves:
// gs_comm.h).  A sidecar is used only when its fingerprints of the two text files still hold (size, modification time, a hash of
// the first and the last MiB); anything else reads as "no sidecar" and the caller parses and builds as before.
// Included at the end of gs_device.hip, after gs_comm.h (it needs TableMeta / table_arrays / table_receive_init).
#pragma once

#include <sys/stat.h>

#include <unistd.h>

#include <cstdio>
#include <thread>

namespace {

struct FileStamp {
    uint64_t size, mtime_ns, head_hash, tail_hash;
};

bool file_stamp(const char *path, FileStamp *out) {
    struct stat st;
    if (stat(path, &st) != 0 || !S_ISREG(st.st_mode)) return false;  // (what can be read only once has no stamp: no sidecar for it)
    out->size = (uint64_t)st.st_size;
    out->mtime_ns = (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec;
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    std::vector<unsigned char> buf(1u << 20);
    auto fnv = [&](size_t n) {
        uint64_t h = 0xCBF29CE484222325ull;
        for (size_t i = 0; i < n; i++) h = (h ^ buf[i]) * 0x100000001B3ull;
        return h;
    };
    size_t n = fread(buf.data(), 1, buf.size(), f);
    out->head_hash = fnv(n);
    out->tail_hash = out->head_hash;
    if (out->size > buf.size()) {
        fseeko(f, (off_t)(out->size - buf.size()), SEEK_SET);
        n = fread(buf.data(), 1, buf.size(), f);
        out->tail_hash = fnv(n);
    }
    fclose(f);
    return true;
}

struct SidecarHeader {
    char magic[8];  // "GSSIDE07"
    uint32_t header_bytes, meta_bytes;
    FileStamp db, loc;
    uint64_t n_rows, names_bytes;
    uint32_t n_contigs, pad;
    TableMeta meta;
};
constexpr char kSidecarMagic[8] = {'G', 'S', 'S', 'I', 'D', 'E', '0', '7'};
constexpr size_t kSidecarChunk = (size_t)64 << 20;

// `bytes` of the file at `off` into dst, read by a few threads (pread: a 5 GiB sidecar comes from the page cache at the speed of
// ONE memcpy otherwise)
bool read_parallel(int fd, void *dst, size_t bytes, uint64_t off) {
    const unsigned T = (unsigned)std::min<size_t>(8, std::max<size_t>(bytes >> 23, 1));
    std::vector<char> okv(T, 1);
    auto work = [&](unsigned t) {
        size_t lo = bytes * t / T, hi = bytes * (t + 1) / T;
        while (lo < hi) {
            const ssize_t r = pread(fd, static_cast<char *>(dst) + lo, hi - lo, (off_t)(off + lo));
            if (r <= 0) {
                okv[t] = 0;
                return;
            }
            lo += (size_t)r;
        }
    };
    gs_parallel_for(T, work);
    for (char o : okv)
        if (!o) return false;
    return true;
}

struct FileCloser {
    FILE *f;
    ~FileCloser() {
        if (f) fclose(f);
    }
};
struct PinnedBuf {
    void *p = nullptr;
    ~PinnedBuf() {
        if (p) (void)hipHostFree(p);
    }
};

}  // namespace

extern "C" {

int gs_sidecar_write(const char *path, const gs_sunkdb *db, const gs_table *t, const char *db_path, const char *loc_path) {
    if (!path || !db || !t || !db_path || !loc_path) {
        set_error("gs_sidecar_write: null argument");
        return GS_E_ARG;
    }
    SidecarHeader h{};
    memcpy(h.magic, kSidecarMagic, 8);
    h.header_bytes = sizeof h;
    h.meta_bytes = sizeof(TableMeta);
    if (!file_stamp(db_path, &h.db) || !file_stamp(loc_path, &h.loc)) {
        set_error("gs_sidecar_write: cannot read %s / %s", db_path, loc_path);
        return GS_E_IO;
    }
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipStreamSynchronize(t->stream));
    int rc;
    uint64_t n_ovf = 0;
    if ((rc = table_n_ovf(t, &n_ovf))) return rc;
    h.meta = table_meta_of(t, n_ovf);
    h.n_rows = db->row_contig.size();
    h.n_contigs = (uint32_t)db->contigs.size();
    std::string names;
    for (const auto &c : db->contigs) {
        names += c;
        names.push_back('\0');
    }
    h.names_bytes = names.size();
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) {
        set_error("gs_sidecar_write: cannot create %s", tmp.c_str());
        return GS_E_IO;
    }
    FileCloser fc{f};
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && (names.empty() || fwrite(names.data(), names.size(), 1, f) == 1);
    auto put = [&](const void *p, size_t bytes) { ok = ok && (!bytes || fwrite(p, bytes, 1, f) == 1); };
    put(db->row_contig.data(), h.n_rows * sizeof(uint32_t));
    put(db->row_start.data(), h.n_rows * sizeof(int64_t));
    put(db->row_group.data(), h.n_rows * sizeof(int64_t));
    PinnedBuf pb;
    HIP_TRY(hipHostMalloc(&pb.p, kSidecarChunk, hipHostMallocDefault));
    gs_table *tm = const_cast<gs_table *>(t);
    for (auto &a : table_arrays(h.meta, tm, t)) {
        const uint8_t *src = static_cast<const uint8_t *>(a.src);
        for (uint64_t o = 0; o < a.bytes && ok; o += kSidecarChunk) {
            const size_t n = (size_t)std::min<uint64_t>(kSidecarChunk, a.bytes - o);
            HIP_TRY(hipMemcpy(pb.p, src + o, n, hipMemcpyDeviceToHost));
            put(pb.p, n);
        }
    }
    fclose(f);
    fc.f = nullptr;
    if (!ok || rename(tmp.c_str(), path) != 0) {
        (void)remove(tmp.c_str());
        set_error("gs_sidecar_write: writing %s failed", path);
        return GS_E_IO;
    }
    return GS_OK;
}

// GS_E_IO: no sidecar, an unreadable one, another format version, or one whose fingerprints no longer match the two files --
// all of them mean "parse and build as usual".
int gs_sidecar_open(const char *path, const char *db_path, const char *loc_path, int device, gs_sunkdb **db_out, gs_table **t_out) {
    if (!path || !db_path || !loc_path || !db_out || !t_out) {
        set_error("gs_sidecar_open: null argument");
        return GS_E_ARG;
    }
    *db_out = nullptr;
    *t_out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) {
        set_error("gs_sidecar_open: no sidecar %s", path);
        return GS_E_IO;
    }
    FileCloser fc{f};
    SidecarHeader h{};
    FileStamp sd{}, sl{};
    if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, kSidecarMagic, 8) != 0 || h.header_bytes != sizeof h ||
        h.meta_bytes != sizeof(TableMeta) || !file_stamp(db_path, &sd) || !file_stamp(loc_path, &sl) ||
        memcmp(&sd, &h.db, sizeof sd) != 0 || memcmp(&sl, &h.loc, sizeof sl) != 0) {
        set_error("gs_sidecar_open: %s is stale or of another version", path);
        return GS_E_IO;
    }
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_sidecar_open: HIP device %d not available -- there is no CPU fallback", device);
        return GS_E_NODEVICE;
    }
    auto db = std::make_unique<gs_sunkdb>();
    db->k = h.meta.k;
    std::string names(h.names_bytes, '\0');
    bool ok = h.names_bytes == 0 || fread(&names[0], h.names_bytes, 1, f) == 1;
    for (size_t o = 0; ok && o < names.size();) {
        const size_t e = names.find('\0', o);
        if (e == std::string::npos) {
            ok = false;
            break;
        }
        db->contigs.emplace_back(names.substr(o, e - o));
        o = e + 1;
    }
    ok = ok && db->contigs.size() == h.n_contigs;
    uint64_t at = ok ? (uint64_t)ftello(f) : 0;
    const int fd = fileno(f);
    auto get = [&](void *p, size_t bytes) {
        ok = ok && (!bytes || read_parallel(fd, p, bytes, at));
        at += bytes;
    };
    if (ok) {
        db->row_contig.resize(h.n_rows);
        db->row_start.resize(h.n_rows);
        db->row_group.resize(h.n_rows);
        get(db->row_contig.data(), h.n_rows * sizeof(uint32_t));
        get(db->row_start.data(), h.n_rows * sizeof(int64_t));
        get(db->row_group.data(), h.n_rows * sizeof(int64_t));
    }
    if (!ok) {
        set_error("gs_sidecar_open: %s is truncated", path);
        return GS_E_IO;
    }
    HIP_TRY(hipSetDevice(device));
    auto *t = new gs_table();
    std::unique_ptr<gs_table, void (*)(gs_table *)> guard(t, gs_table_destroy);
    int rc;
    if ((rc = table_receive_init(t, device, h.meta))) return rc;
    PinnedBuf pb[2];
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; i++) {
        HIP_TRY(hipHostMalloc(&pb[i].p, kSidecarChunk, hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    }
    int cur = 0;
    bool used[2] = {false, false};
    hipError_t he = hipSuccess;
    for (auto &a : table_arrays(h.meta, t, nullptr)) {  // (read of the next chunk overlapped with the upload of this one)
        uint8_t *dst = static_cast<uint8_t *>(*a.dst);
        for (uint64_t o = 0; o < a.bytes && ok && he == hipSuccess; o += kSidecarChunk) {
            const size_t n = (size_t)std::min<uint64_t>(kSidecarChunk, a.bytes - o);
            if (used[cur]) he = hipEventSynchronize(ev[cur]);
            get(pb[cur].p, n);
            if (ok && he == hipSuccess) he = hipMemcpyAsync(dst + o, pb[cur].p, n, hipMemcpyHostToDevice, t->stream);
            if (he == hipSuccess) he = hipEventRecord(ev[cur], t->stream);
            used[cur] = true;
            cur ^= 1;
        }
    }
    if (he == hipSuccess) he = hipStreamSynchronize(t->stream);
    for (auto &e : ev)
        if (e) (void)hipEventDestroy(e);
    if (!ok || he != hipSuccess) {
        set_error("gs_sidecar_open: %s", ok ? hipGetErrorString(he) : "the sidecar is truncated");
        return ok ? GS_E_NODEVICE : GS_E_IO;
    }
    *db_out = db.release();
    *t_out = guard.release();
    return GS_OK;
}

}  // extern "C"
