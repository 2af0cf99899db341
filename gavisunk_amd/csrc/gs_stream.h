// Streaming engine (measurement (ii): host-packed reads -> hits on the host): the main loop of
// workflow/src/kmerpos_annot3.nim:85-96 as a pipeline of read batches against ONE table.  Included at the end of gs_device.hip.
//
//   gs_stream_create   `depth` slots, each a device arena sized ONCE for the largest batch (packed words, per-read and per-tile
//                      arrays, candidate / hit buffers, the pinned mirror of the hit list) + pinned staging for the per-read
//                      index: no allocation, no memset of the whole arena and no synchronous copy per batch
//   gs_stream_submit   host batch: the index (first word / window count / first tile of every read) is computed in one pass
//                      over the lengths into pinned staging; words, index and the tile-descriptor kernel go to the upload
//                      stream; the tagging pass is queued on the table's stream behind an event -- so the upload of batch
//                      i + 1 overlaps the pass of batch i.  `prevLoc` (kmerpos_annot3.nim:82-84) is handed from pass to
//                      pass in device memory (k_chain_carry): no host round trip between batches.
//                      Words in pinned memory (gs_host_alloc) are read by the DMA engine where they lie; pageable words are
//                      staged through the slot's pinned buffer.
//   gs_stream_submit_reads   the same pipeline over batches that already are device-resident (gs_reads handles)
//   gs_stream_next     the OLDEST batch in flight: waits for its pass only, fetches its rows on the copy stream (while the next
//                      pass runs), hands out a view of them that stays valid until the next gs_stream_next
//
// A batch is returned exactly once, in submission order; at most `depth` batches are between submit and next.

struct gs_stream {
    gs_table *t = nullptr;
    int depth = 0;
    uint64_t max_words = 0;
    uint32_t max_reads = 0;
    struct Slot {
        gs_reads *r = nullptr;
        uint64_t *h_start = nullptr;   // pinned: [max_reads + 1]
        uint32_t *h_nwin = nullptr;    // pinned: [max_reads]
        uint32_t *h_tile0 = nullptr;   // pinned: [max_reads]
        uint64_t *h_words = nullptr;   // pinned: [max_words], allocated when the first pageable batch arrives
        hipEvent_t uploaded = nullptr;
        bool busy = false;             // between submit and the gs_stream_next AFTER the one that returned it
    };
    std::vector<Slot> slots;
    struct Entry {
        gs_reads *r;
        int slot;  // -1: a caller-owned batch
    };
    std::vector<Entry> q;     // in flight, oldest first
    Entry held{nullptr, -1};  // returned by the last gs_stream_next
    gs_reads *last = nullptr; // last batch submitted: the next pass takes its carry from it
    uint32_t carry_in = GS_NONE;
    uint32_t carry_out = GS_NONE;
    bool any = false, failed = false;
    hipStream_t h2d = nullptr;
    double probe_ms_seen = 0, total_ms_seen = 0;
};

namespace {

int stream_launch(gs_stream *s, gs_reads *r, int slot) {
    int rc;
    // the predecessor's pass still in flight: its carry is handed over on the device; settled already (or none yet): by value
    const bool chain = s->any && !s->q.empty() && s->q.back().r == s->last;
    const uint32_t carry = !s->any ? s->carry_in : chain ? GS_NONE : s->last->ps.res_carry;
    if ((rc = launch_phase1(s->t, r))) return rc;
    if ((rc = launch_phase2(s->t, r, carry, chain ? s->last : nullptr))) return rc;
    s->any = true;
    s->last = r;
    s->q.push_back(gs_stream::Entry{r, slot});
    return GS_OK;
}

bool is_pinned_host(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary malloc'ed pointer: not an error of ours
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

}  // namespace

extern "C" {

int gs_host_alloc(uint64_t bytes, void **out) {
    if (!out) {
        set_error("gs_host_alloc: bad argument");
        return GS_E_ARG;
    }
    *out = nullptr;
    if (gs_device_count() < 1) {
        set_error("gs_host_alloc: no HIP device");
        return GS_E_NODEVICE;
    }
    HIP_TRY(hipHostMalloc(out, std::max<uint64_t>(bytes, 1), hipHostMallocDefault));
    return GS_OK;
}

void gs_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

void gs_stream_destroy(gs_stream *s) {
    if (!s) return;
    (void)hipSetDevice(s->t->device);
    (void)hipStreamSynchronize(s->t->stream);
    if (s->h2d) (void)hipStreamSynchronize(s->h2d);
    for (auto &sl : s->slots) {
        gs_reads_destroy(sl.r);
        if (sl.h_start) (void)hipHostFree(sl.h_start);
        if (sl.h_nwin) (void)hipHostFree(sl.h_nwin);
        if (sl.h_tile0) (void)hipHostFree(sl.h_tile0);
        if (sl.h_words) (void)hipHostFree(sl.h_words);
        if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
    }
    if (s->h2d) (void)hipStreamDestroy(s->h2d);
    delete s;
}

int gs_stream_create(gs_table *t, uint64_t max_words, uint32_t max_reads, int depth, uint32_t carry_in, gs_stream **out) {
    if (!t || !out || depth < 1 || depth > 16) {
        set_error("gs_stream_create: bad argument (depth 1..16)");
        return GS_E_ARG;
    }
    *out = nullptr;
    HIP_TRY(hipSetDevice(t->device));
    auto *s = new gs_stream();
    std::unique_ptr<gs_stream, void (*)(gs_stream *)> guard(s, gs_stream_destroy);
    s->t = t;
    s->depth = depth;
    s->max_words = max_words;
    s->max_reads = max_reads;
    s->carry_in = s->carry_out = carry_in;
    HIP_TRY(hipStreamCreateWithFlags(&s->h2d, hipStreamNonBlocking));
    if (max_words) {  // (a stream of caller-owned device batches needs no slots)
        // a read of w words has at most w / 64 + 1 tiles (2048 windows = 64 words each)
        const uint64_t cap_tiles = max_words / kTileWords + max_reads + 1;
        s->slots.resize(depth);
        for (auto &sl : s->slots) {
            int rc;
            if ((rc = reads_alloc(t->device, t->k, max_words, max_reads, cap_tiles, (kDirectCand + 4) * cap_tiles + 65536, &sl.r))) return rc;
            HIP_TRY(hipHostMalloc((void **)&sl.h_start, ((uint64_t)max_reads + 1) * sizeof(uint64_t), hipHostMallocDefault));
            HIP_TRY(hipHostMalloc((void **)&sl.h_nwin, std::max<uint64_t>(max_reads, 1) * sizeof(uint32_t), hipHostMallocDefault));
            HIP_TRY(hipHostMalloc((void **)&sl.h_tile0, std::max<uint64_t>(max_reads, 1) * sizeof(uint32_t), hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming));
        }
    }
    *out = guard.release();
    return GS_OK;
}

static int stream_room(gs_stream *s, const char *who) {
    if (!s) {
        set_error("%s: null stream", who);
        return GS_E_ARG;
    }
    if (s->failed) {
        set_error("%s: the stream has stopped at a KeyError (kmerpos_annot3.nim:90 quits there too)", who);
        return GS_E_KEYERROR;
    }
    if ((int)s->q.size() >= s->depth) {
        set_error("%s: %d batches are in flight; take one with gs_stream_next first", who, s->depth);
        return GS_E_CAPACITY;
    }
    return GS_OK;
}

int gs_stream_submit(gs_stream *s, const uint64_t *words, uint64_t n_words, const uint64_t *word_off, const uint32_t *len,
                     uint32_t n_reads) {
    int rc;
    if ((rc = stream_room(s, "gs_stream_submit"))) return rc;
    if ((n_reads && (!word_off || !len)) || (n_words && !words) || n_words > s->max_words || n_reads > s->max_reads) {
        set_error("gs_stream_submit: bad argument, or the batch (%llu words, %u reads) exceeds what the stream was created for (%llu, %u)",
                  (unsigned long long)n_words, n_reads, (unsigned long long)s->max_words, s->max_reads);
        return GS_E_ARG;
    }
    int slot = -1;
    for (int i = 0; i < (int)s->slots.size(); i++)
        if (!s->slots[i].busy) {
            slot = i;
            break;
        }
    if (slot < 0) {  // depth slots, at most depth - 1 in flight + 1 held
        set_error("gs_stream_submit: no free slot (the batch gs_stream_next returned last is still held: call gs_stream_next or gs_stream_release)");
        return GS_E_CAPACITY;
    }
    gs_stream::Slot &sl = s->slots[slot];
    gs_reads *r = sl.r;
    HIP_TRY(hipSetDevice(s->t->device));
    ReadIndexTotals tot;
    if ((rc = read_index_host(s->t->k, word_off, len, n_reads, nullptr, 0, sl.h_start, sl.h_nwin, sl.h_tile0, tot))) return rc;
    if (tot.tiles > s->max_words / kTileWords + s->max_reads + 1) {
        set_error("gs_stream_submit: the batch has more tiles than its words allow (inconsistent lengths)");
        return GS_E_ARG;
    }
    r->n_words = n_words;
    const uint64_t *src = words;
    if (n_words && !is_pinned_host(words)) {  // pageable: through the slot's pinned buffer
        if (!sl.h_words) HIP_TRY(hipHostMalloc((void **)&sl.h_words, std::max<uint64_t>(s->max_words, 1) * sizeof(uint64_t), hipHostMallocDefault));
        memcpy(sl.h_words, words, n_words * sizeof(uint64_t));
        src = sl.h_words;
    }
    if (n_words) HIP_TRY(hipMemcpyAsync(r->words, src, n_words * sizeof(uint64_t), hipMemcpyHostToDevice, s->h2d));
    HIP_TRY(hipMemsetAsync(r->words + n_words, 0, kReadSlackWords * sizeof(uint64_t), s->h2d));
    if ((rc = reads_index_upload(r, sl.h_start, sl.h_nwin, sl.h_tile0, n_reads, tot, s->h2d))) return rc;
    HIP_TRY(hipEventRecord(sl.uploaded, s->h2d));
    HIP_TRY(hipStreamWaitEvent(s->t->stream, sl.uploaded, 0));
    sl.busy = true;
    if ((rc = stream_launch(s, r, slot))) {
        sl.busy = false;
        return rc;
    }
    return GS_OK;
}

int gs_stream_submit_reads(gs_stream *s, gs_reads *r) {
    int rc;
    if ((rc = stream_room(s, "gs_stream_submit_reads"))) return rc;
    if ((rc = check_pair(s->t, r, "gs_stream_submit_reads"))) return rc;
    for (const auto &e : s->q)
        if (e.r == r) {
            set_error("gs_stream_submit_reads: this batch is in flight already");
            return GS_E_ARG;
        }
    if (s->held.r == r) s->held = gs_stream::Entry{nullptr, -1};
    HIP_TRY(hipSetDevice(s->t->device));
    return stream_launch(s, r, -1);
}

int gs_stream_in_flight(const gs_stream *s) { return s ? (int)s->q.size() : 0; }

// lets go of the rows gs_stream_next handed out last (their slot can take the next batch)
void gs_stream_release(gs_stream *s) {
    if (!s) return;
    if (s->held.r && s->held.slot >= 0) s->slots[s->held.slot].busy = false;
    s->held = gs_stream::Entry{nullptr, -1};
}

// The oldest batch in flight.  *hits stays valid until the next gs_stream_next (pinned host memory of the batch's slot).
// GS_E_KEYERROR: the rows before the offending window are handed out and the stream stops.
int gs_stream_next(gs_stream *s, const gs_hit **hits, uint64_t *n_hits, uint32_t *n_reads, gs_counters *ctr) {
    if (!s) {
        set_error("gs_stream_next: null stream");
        return GS_E_ARG;
    }
    if (s->held.r && s->held.slot >= 0) s->slots[s->held.slot].busy = false;
    s->held = gs_stream::Entry{nullptr, -1};
    if (s->q.empty()) {
        set_error("gs_stream_next: no batch in flight");
        return GS_E_ARG;
    }
    const gs_stream::Entry e = s->q.front();
    int rc;
    if ((rc = settle(s->t, e.r))) return rc;
    s->q.erase(s->q.begin());
    s->held = e;
    if (hits) *hits = e.r->hits_host;
    if (n_hits) *n_hits = e.r->ps.res_n;
    if (n_reads) *n_reads = e.r->n_reads;
    s->carry_out = e.r->ps.res_carry;
    if (ctr) {  // of THIS batch (the table's counters accumulate over all passes)
        *ctr = gs_counters{};
        ctr->windows = e.r->n_windows;
        ctr->tiles = e.r->n_tiles;
        ctr->raw_hits = e.r->ps.h_status->totals[0].sum;
        ctr->emitted = e.r->ps.h_status->totals[1].sum;
        ctr->fallback_tiles = e.r->ps.h_status->complex_n;
        ctr->launches = 1;
        ctr->probe_ms = s->t->ctr.probe_ms - s->probe_ms_seen;
        ctr->total_ms = s->t->ctr.total_ms - s->total_ms_seen;
    }
    s->probe_ms_seen = s->t->ctr.probe_ms;
    s->total_ms_seen = s->t->ctr.total_ms;
    if (e.r->ps.res_keyerr) {
        s->failed = true;
        set_error("key not found: a read window is in the SUNK db but has no row in the loc table [KeyError]");
        return GS_E_KEYERROR;
    }
    return GS_OK;
}

// `prevLoc` after the last batch gs_stream_next returned (the carry a later stream, or gs_tag_launch, continues from)
uint32_t gs_stream_carry(const gs_stream *s) { return s ? s->carry_out : GS_NONE; }

}  // extern "C"
