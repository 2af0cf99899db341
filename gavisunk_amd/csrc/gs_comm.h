// Multi-GPU data plane of libgavisunk.so: RCCL over xGMI behind the C-ABI (include/gavisunk.h, gs_comm_*).
// Included at the end of gs_device.hip (it needs the table / reads handles' internals).
//
// The reference has no distributed runtime (its only parallelism is Snakemake's per-chunk scatter, workflow/Snakefile:23-24,
// tagONT.smk:19-37), so this is new design, SURVEY.md section 8(e): reads are sharded over the ranks, the SUNK table is built
// ONCE and replicated, and the path has one exchange step (the 8-byte phase-1 summaries: control plane, the caller's) plus
// the gather of the per-rank hit lists.  Two collectives live here:
//
//   gs_table_broadcast   the BUILT device tables of the root (regions / buckets / payload words / generic slots), array by
//                        array with ncclBroadcast: the other ranks neither parse nor build anything
//   gs_gather_hits       counts and read offsets by ncclAllGather (16 bytes per rank), then grouped ncclSend / ncclRecv of
//                        exactly the rows each rank has -- no pad-to-max copy -- into one device buffer on the root, shifted
//                        to global read indices there and copied to the host once
//
// librccl.so.1 is opened on first use (dlopen): single-GPU runs never load it.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

struct gs_comm {
    int device = 0, nranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    unsigned long long *d_meta = nullptr;  // [4 * nranks]: (count, read offset, status) triples of a gather; summaries; statuses
    gs_hit *d_gather = nullptr;            // root: all ranks' rows
    gs_hit *h_gather = nullptr;            // root: pinned host copy of them (gs_gather_view)
    uint64_t gather_cap = 0, gather_n = 0;
    uint64_t host_cap = 0;    // rows h_gather holds
    bool host_valid = false;  // h_gather holds the rows of the last gather
};

namespace {

struct Rccl {
    void *so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
};

Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        r.so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!r.so) r.so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);  // (never the copy a torch wheel bundles: it is bound to another HIP runtime)
        if (r.so) {
#define GS_RCCL_SYM(name) r.name = reinterpret_cast<decltype(r.name)>(dlsym(r.so, "nccl" #name))
            GS_RCCL_SYM(GetUniqueId);
            GS_RCCL_SYM(CommInitRank);
            GS_RCCL_SYM(CommDestroy);
            GS_RCCL_SYM(GetErrorString);
            GS_RCCL_SYM(Broadcast);
            GS_RCCL_SYM(AllGather);
            GS_RCCL_SYM(AllReduce);
            GS_RCCL_SYM(Send);
            GS_RCCL_SYM(Recv);
            GS_RCCL_SYM(GroupStart);
            GS_RCCL_SYM(GroupEnd);
#undef GS_RCCL_SYM
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GetErrorString || !r.Broadcast || !r.AllGather || !r.AllReduce || !r.Send ||
                !r.Recv || !r.GroupStart || !r.GroupEnd)
                r.so = nullptr;
        }
    }
    if (!r.so) {
        set_error("librccl.so.1 could not be loaded (%s): multi-GPU runs need RCCL", dlerror() ? dlerror() : "missing symbols");
        return nullptr;
    }
    return &r;
}

#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t _r = (expr);                                                                        \
        if (_r != ncclSuccess) {                                                                         \
            set_error("%s failed: %s (%s:%d)", #expr, R->GetErrorString(_r), __FILE__, __LINE__);        \
            return GS_E_NODEVICE;                                                                        \
        }                                                                                                \
    } while (0)

// everything of a table that is not a pointer or a per-handle resource
struct TableMeta {
    int k, shift, compact, c_qb, c_rembits, c_locbits, c_has_flag, c_w64, mz, mz_m, mz_locbits, mz_place, mz_s;
    uint32_t mz_n_regions, join_chunk;
    uint64_t n, cap, c_overflow, mz_overflow, n_ovf, mz_dup, mz_hot;
};

__global__ void k_shift_reads(gs_hit *h, uint64_t n, uint32_t delta) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) h[i].read += delta;
}

// ---- replication of a built table ------------------------------------------------------------------------------------
// ONE piece of code moves a table to another handle, whatever carries the bytes: RCCL broadcasts between ranks
// (gs_table_broadcast), or device-to-device copies inside one GPU (gs_table_clone: the same scalars, the same array list,
// the same receiving-side initialisation -- runnable, and tested, on a one-GPU box).  The receiving side allocates EVERYTHING
// first and the ranks then agree on one status, so that a rank that is out of memory fails all of them instead of leaving
// the others inside a collective.
TableMeta table_meta_of(const gs_table *t, uint64_t n_ovf) {
    return TableMeta{t->k, t->shift, t->compact, t->c_qb, t->c_rembits, t->c_locbits, t->c_has_flag, t->c_w64, t->mz, t->mz_m,
                     t->mz_locbits, t->mz_place, t->mz_s, t->mz_n_regions, 0u, t->n, t->cap, t->c_overflow, t->mz_overflow, n_ovf, t->mz_dup, t->mz_hot};
}

struct TableArray {  // one device array of a table: where its pointer lives in the handle, and its size in bytes
    void **dst;
    const void *src;  // root / clone source: the bytes; receiving rank of a broadcast: NULL
    uint64_t bytes;
};

// the arrays a table with these scalars owns, in a fixed order (both sides derive the same list from the same meta)
std::vector<TableArray> table_arrays(const TableMeta &hm, gs_table *dst, const gs_table *src) {
    std::vector<TableArray> v;
    auto add = [&](void **d, const void *s, uint64_t bytes) {
        if (bytes) v.push_back(TableArray{d, s, bytes});
    };
    add((void **)&dst->slots, src ? src->slots : nullptr, hm.cap * sizeof(Slot));
    if (hm.mz) {
        const uint64_t nslots = (uint64_t)hm.mz_n_regions * kMzWords;
        add((void **)&dst->mz_regions, src ? src->mz_regions : nullptr, nslots * sizeof(unsigned long long));
        add((void **)&dst->mz_pay, src ? src->mz_pay : nullptr, nslots * sizeof(uint32_t));  // (place records: 8 base rows per region)
    } else if (hm.compact) {
        const uint64_t nslots = (uint64_t)(4 * kNPart) << hm.c_qb;
        add((void **)&dst->cbuckets, src ? src->cbuckets : nullptr, (hm.c_w64 ? 2 * nslots : nslots) * sizeof(uint32_t));
        add((void **)&dst->cpayload, src ? src->cpayload : nullptr, nslots * sizeof(uint32_t));
    }
    return v;
}

int table_n_ovf(const gs_table *t, uint64_t *n_ovf) {
    (void)t;
    *n_ovf = 0;  // (the LDS-join layout, retired in round 3, carried a sorted overflow list; the field stays in TableMeta)
    return GS_OK;
}

// receiving side: a fresh handle with the root's scalars and every array allocated (status collected, not returned early)
int table_receive_init(gs_table *t, int device, const TableMeta &hm) {
    int rc;
    if ((rc = table_init(t, device, hm.k, hm.n))) return rc;
    t->shift = hm.shift;
    t->compact = hm.compact;
    t->c_qb = hm.c_qb;
    t->c_rembits = hm.c_rembits;
    t->c_locbits = hm.c_locbits;
    t->c_has_flag = hm.c_has_flag;
    t->c_w64 = hm.c_w64;
    t->mz = hm.mz;
    t->mz_m = hm.mz_m;
    t->mz_locbits = hm.mz_locbits;
    t->mz_place = hm.mz_place;
    t->mz_s = hm.mz_s;
    t->mz_n_regions = hm.mz_n_regions;
    t->cap = hm.cap;
    t->c_overflow = hm.c_overflow;
    t->mz_overflow = hm.mz_overflow;
    t->mz_dup = hm.mz_dup;
    t->mz_hot = hm.mz_hot;
    for (auto &a : table_arrays(hm, t, nullptr)) {
        uint8_t *p = nullptr;
        if ((rc = dev_alloc(&p, a.bytes))) return rc;
        *a.dst = p;
    }
    return GS_OK;
}

// RCCL probes the pointers it is handed with hipPointerGetAttributes, which leaves hipErrorInvalidValue in this thread's last-error
// slot although nothing failed; the next rocprim call (or a HIP_TRY(hipGetLastError()) after one of our launches) would report
// it as its own.  It is dropped AFTER a collective -- that code and only that code.  A sticky error of our own earlier launches
// must still fail the call: the slot is looked at BEFORE the collective, where nothing of RCCL's can be in it.
int comm_enter(const char *who) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: an earlier HIP call on this thread failed: %s", who, hipGetErrorString(e));
        return GS_E_NODEVICE;
    }
    return GS_OK;
}
int comm_leave(const char *who) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && e != hipErrorInvalidValue) {
        set_error("%s: %s", who, hipGetErrorString(e));
        return GS_E_NODEVICE;
    }
    return GS_OK;
}

// every rank contributes a status; all of them get the first non-zero one in rank order (0 = all fine)
int comm_agree(Rccl *R, gs_comm *c, int mine, int *agreed) {
    *agreed = mine;
    if (c->nranks == 1) return GS_OK;
    const unsigned long long v = (unsigned long long)(long long)mine;
    HIP_TRY(hipMemcpyAsync(c->d_meta + c->rank, &v, sizeof v, hipMemcpyHostToDevice, c->stream));
    NCCL_TRY(R->AllGather(c->d_meta + c->rank, c->d_meta, 1, ncclUint64, c->comm, c->stream));
    std::vector<unsigned long long> all((size_t)c->nranks);
    HIP_TRY(hipMemcpyAsync(all.data(), c->d_meta, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (int rc = comm_leave("comm_agree")) return rc;
    *agreed = 0;
    for (int i = 0; i < c->nranks; i++)
        if ((long long)all[i] != 0) {
            *agreed = (int)(long long)all[i];
            break;
        }
    return GS_OK;
}

}  // namespace

extern "C" {

int gs_comm_unique_id(void *id_out) {
    Rccl *R = rccl();
    if (!R) return GS_E_NODEVICE;
    if (!id_out) {
        set_error("gs_comm_unique_id: bad argument");
        return GS_E_ARG;
    }
    ncclUniqueId id;
    NCCL_TRY(R->GetUniqueId(&id));
    static_assert(sizeof id == GS_COMM_ID_BYTES, "GS_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
    memcpy(id_out, &id, sizeof id);
    return GS_OK;
}

void gs_comm_destroy(gs_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    Rccl *R = rccl();
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    (void)hipFree(c->d_meta);
    (void)hipFree(c->d_gather);
    if (c->h_gather) (void)hipHostFree(c->h_gather);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int gs_comm_init(int device, int nranks, int rank, const void *unique_id, gs_comm **out) {
    if (!out || nranks < 1 || rank < 0 || rank >= nranks || !unique_id) {
        set_error("gs_comm_init: bad argument");
        return GS_E_ARG;
    }
    *out = nullptr;
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_comm_init: HIP device %d not available", device);
        return GS_E_NODEVICE;
    }
    Rccl *R = rccl();
    if (!R) return GS_E_NODEVICE;
    HIP_TRY(hipSetDevice(device));
    if (int rc0 = comm_enter("gs_comm_init")) return rc0;
    auto *c = new gs_comm();
    std::unique_ptr<gs_comm, void (*)(gs_comm *)> guard(c, gs_comm_destroy);
    c->device = device;
    c->nranks = nranks;
    c->rank = rank;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    NCCL_TRY(R->CommInitRank(&c->comm, nranks, id, rank));
    int rc;
    if ((rc = dev_alloc(&c->d_meta, (uint64_t)4 * nranks))) return rc;
    if ((rc = comm_leave("gs_comm_init"))) return rc;
    *out = guard.release();
    return GS_OK;
}

int gs_table_broadcast(gs_comm *c, int root, gs_table *mine, gs_table **out) {
    if (!c || !out || root < 0 || root >= c->nranks || (c->rank == root && !mine)) {
        set_error("gs_table_broadcast: bad argument (the root passes its table)");
        return GS_E_ARG;
    }
    *out = nullptr;
    Rccl *R = rccl();
    if (!R) return GS_E_NODEVICE;
    HIP_TRY(hipSetDevice(c->device));
    const bool is_root = c->rank == root;
    if (is_root && mine->device != c->device) {
        set_error("gs_table_broadcast: the table lives on device %d, the communicator on %d", mine->device, c->device);
        return GS_E_ARG;
    }
    if (c->nranks == 1) {
        *out = mine;
        return GS_OK;
    }
    if (int rce = comm_enter("gs_table_broadcast")) return rce;
    // the scalars first (through a device buffer: one collective, no second transport)
    TableMeta hm{};
    int rc = GS_OK;
    if (is_root) {
        HIP_TRY(hipStreamSynchronize(mine->stream));
        uint64_t n_ovf = 0;
        if ((rc = table_n_ovf(mine, &n_ovf))) return rc;
        hm = table_meta_of(mine, n_ovf);
    }
    TableMeta *dm = nullptr;
    if ((rc = dev_alloc(&dm, 1))) return rc;
    DevBufs tmp;
    tmp.p.push_back(dm);
    if (is_root) HIP_TRY(hipMemcpy(dm, &hm, sizeof hm, hipMemcpyHostToDevice));
    NCCL_TRY(R->Broadcast(dm, dm, sizeof hm, ncclUint8, root, c->comm, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!is_root) HIP_TRY(hipMemcpy(&hm, dm, sizeof hm, hipMemcpyDeviceToHost));
    gs_table *t = mine;
    std::unique_ptr<gs_table, void (*)(gs_table *)> guard(nullptr, gs_table_destroy);
    int st = GS_OK;
    if (!is_root) {
        t = new gs_table();
        guard.reset(t);
        st = table_receive_init(t, c->device, hm);  // every allocation of this rank, BEFORE any array collective
    }
    int agreed = 0;
    if ((rc = comm_agree(R, c, st, &agreed))) return rc;
    if (agreed) {
        if (!st) set_error("gs_table_broadcast: another rank could not receive the table (status %d)", agreed);
        return agreed;
    }
    // the arrays, one broadcast each, in pieces of 1 GiB (config 3's region array is 74 GiB: no single collective of that size)
    for (auto &a : table_arrays(hm, t, is_root ? mine : nullptr)) {
        uint8_t *b = reinterpret_cast<uint8_t *>(*a.dst);
        const size_t piece = (size_t)1 << 30;
        for (size_t o = 0; o < a.bytes; o += piece)
            NCCL_TRY(R->Broadcast(b + o, b + o, std::min<size_t>(piece, a.bytes - o), ncclUint8, root, c->comm, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (int rcl = comm_leave("gs_table_broadcast")) return rcl;
    if (!is_root) guard.release();
    *out = t;
    return GS_OK;
}

// The same replication with device-to-device copies instead of broadcasts: a second handle of the table on ITS device.  What
// a receiving rank of gs_table_broadcast does -- scalars through TableMeta, every array allocated from them, the handle
// initialised without a build -- runs here on one GPU.
int gs_table_clone(const gs_table *src, gs_table **out) {
    if (!src || !out) {
        set_error("gs_table_clone: bad argument");
        return GS_E_ARG;
    }
    *out = nullptr;
    HIP_TRY(hipSetDevice(src->device));
    HIP_TRY(hipStreamSynchronize(src->stream));
    int rc;
    uint64_t n_ovf = 0;
    if ((rc = table_n_ovf(src, &n_ovf))) return rc;
    const TableMeta sent = table_meta_of(src, n_ovf);
    // (the scalars take the route they take between ranks: packed, through device memory, unpacked)
    TableMeta *dm = nullptr;
    if ((rc = dev_alloc(&dm, 1))) return rc;
    DevBufs tmp;
    tmp.p.push_back(dm);
    TableMeta hm{};
    HIP_TRY(hipMemcpy(dm, &sent, sizeof sent, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(&hm, dm, sizeof hm, hipMemcpyDeviceToHost));
    auto *t = new gs_table();
    std::unique_ptr<gs_table, void (*)(gs_table *)> guard(t, gs_table_destroy);
    if ((rc = table_receive_init(t, src->device, hm))) return rc;
    for (auto &a : table_arrays(hm, t, src)) HIP_TRY(hipMemcpyAsync(*a.dst, a.src, a.bytes, hipMemcpyDeviceToDevice, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    *out = guard.release();
    return GS_OK;
}

// ncclAllGather of n 64-bit words per rank on the library's stream (the phase-1 summaries of a read-sharded pass: SURVEY.md
// section 8e); out[rank * n + i] on the host
int gs_comm_allgather_u64(gs_comm *c, const uint64_t *mine, uint32_t n, uint64_t *out) {
    if (!c || !mine || !out || !n || n > 2) {
        set_error("gs_comm_allgather_u64: bad argument (1 or 2 words per rank)");
        return GS_E_ARG;
    }
    Rccl *R = rccl();
    if (!R) return GS_E_NODEVICE;
    HIP_TRY(hipSetDevice(c->device));
    if (int rce = comm_enter("gs_comm_allgather_u64")) return rce;
    unsigned long long *seg = c->d_meta + (size_t)n * c->rank;
    HIP_TRY(hipMemcpyAsync(seg, mine, n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    if (c->nranks > 1) NCCL_TRY(R->AllGather(seg, c->d_meta, n, ncclUint64, c->comm, c->stream));
    HIP_TRY(hipMemcpyAsync(out, c->d_meta, (size_t)n * c->nranks * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return comm_leave("gs_comm_allgather_u64");
}

int gs_gather_hits(gs_comm *c, int root, gs_table *t, gs_reads *r, uint32_t read_offset, gs_hit *out, uint64_t cap, uint64_t *n_out) {
    int rc;
    if (!c || root < 0 || root >= c->nranks) {
        set_error("gs_gather_hits: bad argument");
        return GS_E_ARG;
    }
    Rccl *R = rccl();
    if (!R) return GS_E_NODEVICE;
    HIP_TRY(hipSetDevice(c->device));
    // A rank whose pass failed -- a KeyError is data-dependent: one shard meets the db k-mer without loc row, the others do not
    // (kmerpos_annot3.nim:90) -- must not leave the others inside a collective: its status travels WITH the counts, every rank
    // takes part in the exchange of the meta, and all of them return the same code before any row is sent.
    // (a sticky error of an earlier launch of ours counts as this rank's failure and travels with the counts as well)
    int st = comm_enter("gs_gather_hits");
    if (!st) st = check_pair(t, r, "gs_gather_hits");
    if (!st) st = settle(t, r);
    if (!st && r->ps.res_keyerr) {
        set_error("key not found: a read window is in the SUNK db but has no row in the loc table [KeyError]");
        st = GS_E_KEYERROR;
    }
    const bool is_root = c->rank == root;
    const int n = c->nranks;
    // (count, read offset, status) of every rank
    const unsigned long long mine[3] = {st ? 0ull : (unsigned long long)r->ps.res_n, read_offset, (unsigned long long)(long long)st};
    HIP_TRY(hipMemcpyAsync(c->d_meta + 3 * c->rank, mine, sizeof mine, hipMemcpyHostToDevice, c->stream));
    if (n > 1) NCCL_TRY(R->AllGather(c->d_meta + 3 * c->rank, c->d_meta, 3, ncclUint64, c->comm, c->stream));
    std::vector<unsigned long long> meta((size_t)3 * n);
    HIP_TRY(hipMemcpyAsync(meta.data(), c->d_meta, meta.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint64_t total = 0;
    for (int i = 0; i < n; i++) total += meta[3 * i];
    for (int i = 0; i < n; i++)
        if ((long long)meta[3 * i + 2] != 0) {
            const int code = (int)(long long)meta[3 * i + 2];
            if (i != c->rank) set_error("gs_gather_hits: the pass of rank %d failed (status %d%s)", i, code, code == GS_E_KEYERROR ? ": KeyError" : "");
            if (n_out) *n_out = 0;
            return code;
        }
    if (n_out) *n_out = total;
    // the root's buffer, then one more agreement: the others only send when the root can receive
    st = GS_OK;
    if (is_root) {
        if (c->gather_cap < total) {  // grown once; later passes reuse the buffer
            (void)hipFree(c->d_gather);
            c->d_gather = nullptr;
            c->gather_cap = 0;
            const uint64_t want = total + (total >> 3) + 1024;
            if ((st = dev_alloc(&c->d_gather, want)) == GS_OK) c->gather_cap = want;
        }
        c->gather_n = st ? 0 : total;
        c->host_valid = false;
    }
    int agreed = 0;
    if ((rc = comm_agree(R, c, st, &agreed))) return rc;
    if (agreed) {
        if (!st) set_error("gs_gather_hits: the root could not allocate the gather buffer (status %d)", agreed);
        return agreed;
    }
    // rows: every rank sends exactly what it has; the root receives them back to back in rank order
    if (n > 1) NCCL_TRY(R->GroupStart());
    if (is_root) {
        uint64_t at = 0;
        for (int i = 0; i < n; i++) {
            const uint64_t cnt = meta[3 * i];
            if (i == root) {
                if (cnt) HIP_TRY(hipMemcpyAsync(c->d_gather + at, r->hits, cnt * sizeof(gs_hit), hipMemcpyDeviceToDevice, c->stream));
            } else if (cnt) {
                NCCL_TRY(R->Recv(c->d_gather + at, (size_t)cnt * sizeof(gs_hit), ncclUint8, i, c->comm, c->stream));
            }
            at += cnt;
        }
    } else if (r->ps.res_n) {
        NCCL_TRY(R->Send(r->hits, (size_t)r->ps.res_n * sizeof(gs_hit), ncclUint8, root, c->comm, c->stream));
    }
    if (n > 1) NCCL_TRY(R->GroupEnd());
    // (RCCL probes pointers with hipPointerGetAttributes and leaves "invalid argument" in this thread's last-error slot: that code
    // is dropped here -- comm_leave -- so that the check after OUR kernels below, and rocprim's own checks in later calls, see only
    // our errors)
    if ((rc = comm_leave("gs_gather_hits"))) return rc;
    if (is_root) {
        uint64_t at = 0;
        for (int i = 0; i < n; i++) {  // shard-local read indices -> indices in the whole input
            const uint64_t cnt = meta[3 * i];
            if (cnt && meta[3 * i + 1])
                hipLaunchKernelGGL(k_shift_reads, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, c->d_gather + at, cnt,
                                   (uint32_t)meta[3 * i + 1]);
            at += cnt;
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    // the concatenated rows now are in the root's device memory (gs_gather_device); they cross PCIe only when asked for: into
    // `out` here, or into pinned host memory by gs_gather_view (every rank's own rows already are on its own host after
    // gs_tag_finish -- eight PCIe links in parallel -- so a pipeline that writes per-shard output never needs this copy)
    if (is_root && out && total > cap) {
        set_error("%llu hits do not fit in cap = %llu", (unsigned long long)total, (unsigned long long)cap);
        return GS_E_CAPACITY;
    }
    if (is_root && out && total) HIP_TRY(hipMemcpy(out, c->d_gather, total * sizeof(gs_hit), hipMemcpyDeviceToHost));
    return GS_OK;
}

int gs_gather_view(gs_comm *c, const gs_hit **out, uint64_t *n_out) {
    if (!c) {
        set_error("gs_gather_view: bad argument");
        return GS_E_ARG;
    }
    if (n_out) *n_out = c->gather_n;
    if (out) {
        *out = nullptr;
        if (c->gather_n) {
            HIP_TRY(hipSetDevice(c->device));
            if (c->host_cap < c->gather_n) {
                if (c->h_gather) (void)hipHostFree(c->h_gather);
                c->h_gather = nullptr;
                c->host_cap = 0;
                const uint64_t want = c->gather_n + (c->gather_n >> 3) + 1024;
                HIP_TRY(hipHostMalloc((void **)&c->h_gather, (size_t)want * sizeof(gs_hit), hipHostMallocDefault));
                c->host_cap = want;
                c->host_valid = false;
            }
            if (!c->host_valid) {
                HIP_TRY(hipMemcpy(c->h_gather, c->d_gather, c->gather_n * sizeof(gs_hit), hipMemcpyDeviceToHost));
                c->host_valid = true;
            }
            *out = c->h_gather;
        }
    }
    return GS_OK;
}

int gs_gather_device(gs_comm *c, const void **d_out, uint64_t *n_out) {
    if (!c) {
        set_error("gs_gather_device: bad argument");
        return GS_E_ARG;
    }
    if (d_out) *d_out = c->d_gather;
    if (n_out) *n_out = c->gather_n;
    return GS_OK;
}

// ---- the second half of a read-sharded run (csrc/gs_chain.hip; SURVEY.md section 8e: "pair-distance stage: shard by read") ----
// Every rank filters and pair-tests its OWN rows; two things are not local.  The bad-SUNK histogram counts reads per SUNK over
// all reads (badsunks_AR.py:20-46): one in-place sum of the per-key counters (n_keys x 4 bytes) per haplotype.  The contig-wide
// union joins the reads' ID sets (process-by-contig_lowmem_AR.py:219-260): the TSV rows of every rank -- 16 bytes per row, 10^-2
// of the hit list -- go to the root, which merges them (gs_pairstage_from_rows).  The hit lists themselves never leave their GPU.
int gs_comm_allreduce_u32(gs_comm *c, void *d_buf, uint64_t n) {
    if (!c || (n && !d_buf)) {
        set_error("gs_comm_allreduce_u32: bad argument");
        return GS_E_ARG;
    }
    Rccl *R = rccl();
    if (!R) return GS_E_NODEVICE;
    HIP_TRY(hipSetDevice(c->device));
    // (a failure here leaves the other ranks in the collective: callers agree on a status BEFORE they come here -- dist.py)
    if (int rce = comm_enter("gs_comm_allreduce_u32")) return rce;
    const uint64_t piece = (uint64_t)1 << 28;  // 1 GiB of counters per collective
    for (uint64_t o = 0; o < n; o += piece) {
        uint32_t *b = static_cast<uint32_t *>(d_buf) + o;
        NCCL_TRY(R->AllReduce(b, b, (size_t)std::min(piece, n - o), ncclUint32, ncclSum, c->comm, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return comm_leave("gs_comm_allreduce_u32");
}
// the same as a gs_reduce_u32_fn (gs_rowset_bad_keys_reduced): ctx = the communicator
int gs_comm_reduce_u32(void *ctx, void *d_buf, uint64_t n) { return gs_comm_allreduce_u32(static_cast<gs_comm *>(ctx), d_buf, n); }

// Variable-size gather of host bytes to the root: (size, status) of every rank by ncclAllGather -- a rank whose step failed says
// so here and every rank returns its code before anything is sent --, then grouped ncclSend / ncclRecv of exactly the bytes
// each rank has, staged through device memory on both sides.  sizes[nranks] on every rank; *out (root only; pinned: gs_host_free) = the
// ranks' bytes back to back in rank order.
int gs_comm_gatherv_host(gs_comm *c, int root, const void *mine, uint64_t nbytes, int status, void **out, uint64_t *sizes) {
    if (!c || root < 0 || root >= c->nranks || (nbytes && !mine) || !sizes || (c->rank == root && !out)) {
        set_error("gs_comm_gatherv_host: bad argument");
        return GS_E_ARG;
    }
    if (out) *out = nullptr;
    Rccl *R = rccl();
    if (!R) return GS_E_NODEVICE;
    HIP_TRY(hipSetDevice(c->device));
    const int n = c->nranks;
    const bool is_root = c->rank == root;
    if (!status) status = comm_enter("gs_comm_gatherv_host");  // (a sticky error of ours travels as this rank's status: nobody is left waiting)
    else (void)hipGetLastError();
    const unsigned long long meta_mine[2] = {status ? 0ull : (unsigned long long)nbytes, (unsigned long long)(long long)status};
    HIP_TRY(hipMemcpyAsync(c->d_meta + 2 * c->rank, meta_mine, sizeof meta_mine, hipMemcpyHostToDevice, c->stream));
    if (n > 1) NCCL_TRY(R->AllGather(c->d_meta + 2 * c->rank, c->d_meta, 2, ncclUint64, c->comm, c->stream));
    std::vector<unsigned long long> meta((size_t)2 * n);
    HIP_TRY(hipMemcpyAsync(meta.data(), c->d_meta, meta.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint64_t total = 0;
    for (int i = 0; i < n; i++) {
        sizes[i] = meta[2 * i];
        total += meta[2 * i];
    }
    for (int i = 0; i < n; i++)
        if ((long long)meta[2 * i + 1] != 0) {
            const int code = (int)(long long)meta[2 * i + 1];
            if (i != c->rank) set_error("gs_comm_gatherv_host: the step of rank %d failed (status %d)", i, code);
            return code;
        }
    // staging buffers, then one more agreement (nobody sends to a root that cannot receive)
    const uint64_t my_bytes = meta[2 * c->rank];
    uint8_t *d_stage = nullptr;
    int st = dev_alloc(&d_stage, std::max<uint64_t>(is_root ? total : my_bytes, 16));
    DevBufs tmp;
    if (!st) tmp.p.push_back(d_stage);
    void *host = nullptr;
    if (!st && is_root && hipHostMalloc(&host, std::max<uint64_t>(total, 1), hipHostMallocDefault) != hipSuccess) {
        set_error("gs_comm_gatherv_host: out of pinned host memory (%llu bytes)", (unsigned long long)total);
        host = nullptr;
        st = GS_E_NOMEM;
    }
    int agreed = 0, rc;
    if ((rc = comm_agree(R, c, st, &agreed))) {
        gs_host_free(host);
        return rc;
    }
    if (agreed) {
        if (!st) set_error("gs_comm_gatherv_host: another rank could not allocate its buffer (status %d)", agreed);
        gs_host_free(host);
        return agreed;
    }
    uint64_t my_at = 0;
    for (int i = 0; i < c->rank; i++) my_at += meta[2 * i];
    uint8_t *d_mine = is_root ? d_stage + my_at : d_stage;
    hipError_t e = my_bytes ? hipMemcpyAsync(d_mine, mine, my_bytes, hipMemcpyHostToDevice, c->stream) : hipSuccess;
    ncclResult_t nr = ncclSuccess;
    if (e == hipSuccess && n > 1) {
        nr = R->GroupStart();
        if (is_root) {
            uint64_t at = 0;
            for (int i = 0; i < n && nr == ncclSuccess; i++) {
                if (i != root && meta[2 * i]) nr = R->Recv(d_stage + at, (size_t)meta[2 * i], ncclUint8, i, c->comm, c->stream);
                at += meta[2 * i];
            }
        } else if (my_bytes && nr == ncclSuccess) {
            nr = R->Send(d_stage, (size_t)my_bytes, ncclUint8, root, c->comm, c->stream);
        }
        const ncclResult_t ge = R->GroupEnd();
        if (nr == ncclSuccess) nr = ge;
    }
    if (e == hipSuccess && nr == ncclSuccess && is_root && total) e = hipMemcpyAsync(host, d_stage, total, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess || nr != ncclSuccess) {
        set_error("gs_comm_gatherv_host: %s", e != hipSuccess ? hipGetErrorString(e) : R->GetErrorString(nr));
        gs_host_free(host);
        return GS_E_NODEVICE;
    }
    if (is_root) *out = host;
    return comm_leave("gs_comm_gatherv_host");
}

// plain copies between host memory and device pointers this library handed out (the counters of gs_rowset_bad_keys_reduced when
// the caller's data plane is not RCCL: tests with several processes on one GPU)
int gs_device_copy_to_host(int device, void *host, const void *d_src, uint64_t nbytes) {
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_device_copy_to_host: HIP device %d not available", device);
        return GS_E_NODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    if (nbytes) HIP_TRY(hipMemcpy(host, d_src, nbytes, hipMemcpyDeviceToHost));
    return GS_OK;
}
int gs_device_copy_from_host(int device, void *d_dst, const void *host, uint64_t nbytes) {
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_device_copy_from_host: HIP device %d not available", device);
        return GS_E_NODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    if (nbytes) HIP_TRY(hipMemcpy(d_dst, host, nbytes, hipMemcpyHostToDevice));
    return GS_OK;
}

}  // extern "C"
