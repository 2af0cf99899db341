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
    unsigned long long *d_meta = nullptr;  // [2 * nranks] (count, read offset) pairs
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
            GS_RCCL_SYM(Send);
            GS_RCCL_SYM(Recv);
            GS_RCCL_SYM(GroupStart);
            GS_RCCL_SYM(GroupEnd);
#undef GS_RCCL_SYM
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GetErrorString || !r.Broadcast || !r.AllGather || !r.Send ||
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
    int k, shift, compact, c_qb, c_rembits, c_locbits, c_has_flag, c_w64, mz, mz_m, mz_locbits, join;
    uint32_t mz_n_regions, join_chunk;
    uint64_t n, cap, c_overflow, mz_overflow, n_ovf;
};

__global__ void k_shift_reads(gs_hit *h, uint64_t n, uint32_t delta) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) h[i].read += delta;
}

// one device array: root sends, the others allocate and receive
template <typename T>
int bcast_array(Rccl *R, gs_comm *c, int root, T **p, uint64_t count) {
    if (!count) return GS_OK;
    int rc;
    if (c->rank != root && (rc = dev_alloc(p, count))) return rc;
    // in pieces of 1 GiB (config 3's region array is 74 GiB: no single collective of that size)
    const size_t bytes = (size_t)count * sizeof(T), piece = (size_t)1 << 30;
    uint8_t *b = reinterpret_cast<uint8_t *>(*p);
    for (size_t o = 0; o < bytes; o += piece) NCCL_TRY(R->Broadcast(b + o, b + o, std::min(piece, bytes - o), ncclUint8, root, c->comm, c->stream));
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
    if ((rc = dev_alloc(&c->d_meta, (uint64_t)2 * nranks))) return rc;
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
    // the scalars first (through a device buffer: one collective, no second transport)
    TableMeta hm{};
    uint64_t n_ovf = 0;
    if (is_root) {
        gs_table *t = mine;
        HIP_TRY(hipStreamSynchronize(t->stream));
        if (t->join) {
            uint32_t last = 0;
            HIP_TRY(hipMemcpy(&last, t->j_ovf_start + kJoinSlices, sizeof last, hipMemcpyDeviceToHost));
            n_ovf = last;
        }
        hm = TableMeta{t->k, t->shift, t->compact, t->c_qb, t->c_rembits, t->c_locbits, t->c_has_flag, t->c_w64, t->mz, t->mz_m,
                       t->mz_locbits, t->join, t->mz_n_regions, t->join_chunk, t->n, t->cap, t->c_overflow, t->mz_overflow, n_ovf};
    }
    TableMeta *dm = nullptr;
    int rc;
    if ((rc = dev_alloc(&dm, 1))) return rc;
    DevBufs tmp;
    tmp.p.push_back(dm);
    if (is_root) HIP_TRY(hipMemcpy(dm, &hm, sizeof hm, hipMemcpyHostToDevice));
    NCCL_TRY(R->Broadcast(dm, dm, sizeof hm, ncclUint8, root, c->comm, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!is_root) HIP_TRY(hipMemcpy(&hm, dm, sizeof hm, hipMemcpyDeviceToHost));
    gs_table *t = mine;
    std::unique_ptr<gs_table, void (*)(gs_table *)> guard(nullptr, gs_table_destroy);
    if (!is_root) {
        t = new gs_table();
        guard.reset(t);
        if ((rc = table_init(t, c->device, hm.k, hm.n))) return rc;
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
        t->join = hm.join;
        t->mz_n_regions = hm.mz_n_regions;
        t->join_chunk = hm.join_chunk;
        t->cap = hm.cap;
        t->c_overflow = hm.c_overflow;
        t->mz_overflow = hm.mz_overflow;
    }
    // the arrays, one broadcast each
    if ((rc = bcast_array(R, c, root, &t->slots, hm.cap))) return rc;
    if (hm.mz) {
        const uint64_t nslots = (uint64_t)hm.mz_n_regions * kMzWords;
        if ((rc = bcast_array(R, c, root, reinterpret_cast<unsigned long long **>(&t->mz_regions), nslots)) ||
            (rc = bcast_array(R, c, root, &t->mz_pay, nslots)))
            return rc;
    } else if (hm.compact) {
        const uint64_t nslots = (uint64_t)(4 * kNPart) << hm.c_qb;
        if ((rc = bcast_array(R, c, root, reinterpret_cast<uint32_t **>(&t->cbuckets), hm.c_w64 ? 2 * nslots : nslots)) ||
            (rc = bcast_array(R, c, root, &t->cpayload, nslots)))
            return rc;
        if (hm.join) {
            if ((rc = bcast_array(R, c, root, &t->j_ovf_start, (uint64_t)kJoinSlices + 1))) return rc;
            if (!is_root && (rc = dev_alloc(&t->j_ovf_code, hm.n_ovf)) == GS_OK) rc = dev_alloc(&t->j_ovf_loc, hm.n_ovf);
            if (rc) return rc;
            if (hm.n_ovf) {  // (at most a few thousand keys per slice: small)
                NCCL_TRY(R->Broadcast(t->j_ovf_code, t->j_ovf_code, (size_t)hm.n_ovf * sizeof(uint64_t), ncclUint8, root, c->comm, c->stream));
                NCCL_TRY(R->Broadcast(t->j_ovf_loc, t->j_ovf_loc, (size_t)hm.n_ovf * sizeof(uint32_t), ncclUint8, root, c->comm, c->stream));
            }
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!is_root) guard.release();
    *out = t;
    return GS_OK;
}

int gs_gather_hits(gs_comm *c, int root, gs_table *t, gs_reads *r, uint32_t read_offset, gs_hit *out, uint64_t cap, uint64_t *n_out) {
    int rc;
    if (!c || root < 0 || root >= c->nranks) {
        set_error("gs_gather_hits: bad argument");
        return GS_E_ARG;
    }
    if ((rc = check_pair(t, r, "gs_gather_hits"))) return rc;
    if ((rc = settle(t, r))) return rc;
    if (r->ps.res_keyerr) {
        set_error("key not found: a read window is in the SUNK db but has no row in the loc table [KeyError]");
        return GS_E_KEYERROR;
    }
    Rccl *R = rccl();
    if (!R) return GS_E_NODEVICE;
    HIP_TRY(hipSetDevice(c->device));
    const bool is_root = c->rank == root;
    const int n = c->nranks;
    // (count, read offset) of every rank
    const unsigned long long mine[2] = {r->ps.res_n, read_offset};
    HIP_TRY(hipMemcpyAsync(c->d_meta + 2 * c->rank, mine, sizeof mine, hipMemcpyHostToDevice, c->stream));
    if (n > 1) NCCL_TRY(R->AllGather(c->d_meta + 2 * c->rank, c->d_meta, 2, ncclUint64, c->comm, c->stream));
    std::vector<unsigned long long> meta((size_t)2 * n);
    HIP_TRY(hipMemcpyAsync(meta.data(), c->d_meta, meta.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint64_t total = 0;
    for (int i = 0; i < n; i++) total += meta[2 * i];
    if (n_out) *n_out = total;
    if (is_root) {
        if (out && total > cap) {
            set_error("%llu hits do not fit in cap = %llu", (unsigned long long)total, (unsigned long long)cap);
            // the other ranks are about to send: take part in the exchange below with a scratch buffer, then report
        }
        if (c->gather_cap < total) {  // grown once; later passes reuse the buffer
            (void)hipFree(c->d_gather);
            c->d_gather = nullptr;
            c->gather_cap = 0;
            const uint64_t want = total + (total >> 3) + 1024;
            if ((rc = dev_alloc(&c->d_gather, want))) return rc;
            c->gather_cap = want;
        }
        c->gather_n = total;
        c->host_valid = false;
    }
    // rows: every rank sends exactly what it has; the root receives them back to back in rank order
    if (n > 1) NCCL_TRY(R->GroupStart());
    if (is_root) {
        uint64_t at = 0;
        for (int i = 0; i < n; i++) {
            const uint64_t cnt = meta[2 * i];
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
    if (is_root) {
        uint64_t at = 0;
        for (int i = 0; i < n; i++) {  // shard-local read indices -> indices in the whole input
            const uint64_t cnt = meta[2 * i];
            if (cnt && meta[2 * i + 1])
                hipLaunchKernelGGL(k_shift_reads, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, c->d_gather + at, cnt,
                                   (uint32_t)meta[2 * i + 1]);
            at += cnt;
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    // the concatenated rows now are in the root's device memory (gs_gather_device); they cross PCIe only when asked for: into
    // `out` here, or into pinned host memory by gs_gather_view (every rank's own rows already are on its own host after
    // gs_tag_finish -- eight PCIe links in parallel -- so a pipeline that writes per-shard output never needs this copy)
    if (is_root && out && total && total <= cap) HIP_TRY(hipMemcpy(out, c->d_gather, total * sizeof(gs_hit), hipMemcpyDeviceToHost));
    if (is_root && out && total > cap) return GS_E_CAPACITY;
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

}  // extern "C"
