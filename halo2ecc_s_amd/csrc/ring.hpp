// h2e_ring: output arrays for MORE runs in flight than full array sets fit the device.
// Included by h2e_capi.cpp inside its extern "C" block, behind run.hpp.
//
// A pipelined run holds its output arrays from its first kernel to its completion, and its value chain has to be finished before its
// big expansion can stream: with two array sets (2 x 110 GB of 288 for 64 x 1024-point MSM tiles) the step is half a run's latency -
// chain + expansion + tail - not the expansions' 12-13 ms, and a third set does not fit.  But only ONE launch of such a program is big:
// the MSM's window strands own 81-89 % of the rows of every array, and nothing writes those rows before the windows' value chain stores
// its operand cells - 2/3 into the run.  So the rows of the program's biggest launch (a contiguous row range of each array) are backed by
// TWO physical copies and everything else by `depth` of them, mapped into `depth`-many... lcm(2, depth) virtual array sets with the HIP
// virtual-memory API (one physical piece at several addresses; exp/ubench/vmm_alias.hip: supported, 4 KB granularity, full rate):
// run k writes set k mod lcm = (rest copy k mod depth, big-launch copy k mod 2).  Three runs in flight cost 2.2 sets of memory.
// What makes it safe is one more edge in the run's schedule: everything of run k that WRITES the big launch's rows (its value chain's
// stores, its expansion, its fix-ups) waits for the completion of run k - 2, the previous user of the same physical rows.
struct h2e_ring {
    h2e_ctx* ctx = nullptr;
    h2e_program* prog = nullptr;
    uint32_t n_instances = 0, depth = 0, n_virtual = 0;
    int big_seg = -1, big_launch = -1;
    struct Piece { hipMemGenericAllocationHandle_t h; size_t bytes; };
    struct Arr {
        size_t total = 0, a0 = 0, a1 = 0;          // bytes (granule multiples): [a0, a1) = the big launch's rows, shrunk to whole granules
        std::vector<Piece> big[2];                   // physical copies of [a0, a1)
        std::vector<std::vector<Piece>> rest, rest2; // [depth] physical copies of [0, a0) and of [a1, total) (hipMemMap takes whole pieces only)
        std::vector<void*> va;                       // [n_virtual]
        uint64_t row0 = 0, row1 = 0;                 // the big launch's rows
    } arr[3];
    std::vector<int> jobs;                           // job of run k mod depth
    std::vector<hipEvent_t> released;                // [depth]: recorded by h2e_ring_release(k) on the consumer's stream
    std::vector<uint64_t> released_k;                // ... for which run (~0: none)
    uint64_t next = 0;
    size_t gran = 0;
};

static void ring_free(h2e_ring* r) {
    if (!r) return;
    (void)hipSetDevice(r->ctx->device);
    (void)hipDeviceSynchronize();
    for (hipEvent_t e : r->released)
        if (e) (void)hipEventDestroy(e);
    for (auto& a : r->arr) {
        for (void* va : a.va)
            if (va) {
                (void)hipMemUnmap(va, a.total);
                (void)hipMemAddressFree(va, a.total);
            }
        for (auto& v : a.big)
            for (auto& pc : v) (void)hipMemRelease(pc.h);
        for (auto& v : a.rest)
            for (auto& pc : v) (void)hipMemRelease(pc.h);
        for (auto& v : a.rest2)
            for (auto& pc : v) (void)hipMemRelease(pc.h);
    }
    delete r;
}
void h2e_ring_destroy(h2e_ring* r) { ring_free(r); }

int h2e_ring_create(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, uint32_t depth, h2e_ring** out) {
    if (!ctx || !p || !out) return fail(H2E_ERR_INVALID, "null argument");
    *out = nullptr;
    if (n_instances == 0 || depth < 2 || depth > (uint32_t)h2e_ctx::N_SLOTS) return fail(H2E_ERR_INVALID, "h2e_ring_create: depth must be 2 .. 32, n_instances > 0");
    std::lock_guard<std::mutex> guard(ctx->mu);
    if (ctx->depth != depth) return fail(H2E_ERR_INVALID, "h2e_ring_create: set H2E_OPT_PIPELINE_DEPTH to the ring's depth first");
    HIP_TRY(hipSetDevice(ctx->device));
    const h2e::Recorder& rec = *p->rec;
    // the biggest launch by rows: a fork (its strands' rows are one contiguous range of every array)
    int big = -1, launch = -1, li = 0;
    uint64_t best = 0;
    for (size_t si = 0; si < rec.segments.size(); si++) {
        const h2e::Segment& s = rec.segments[si];
        if (s.tape_end <= s.tape_begin) continue;
        uint64_t rows = s.is_fork ? (uint64_t)s.n_strands * ((uint64_t)s.dbase + s.drange + s.dselect) : 0;
        if (rows > best) { best = rows; big = (int)si; launch = li; }
        li++;
    }
    if (big < 0) return fail(H2E_ERR_INVALID, "h2e_ring_create: the program has no forked launch whose rows could be shared");
    const h2e::Segment& bs = rec.segments[big];
    std::unique_ptr<h2e_ring, void (*)(h2e_ring*)> r(new h2e_ring, ring_free);
    r->ctx = ctx; r->prog = p; r->n_instances = n_instances; r->depth = depth;
    r->n_virtual = depth % 2 == 0 ? depth : 2 * depth;
    r->big_seg = big; r->big_launch = launch;
    r->jobs.assign(depth, -1);
    r->released.assign(depth, nullptr);
    r->released_k.assign(depth, ~0ull);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = ctx->device;
    HIP_TRY(hipMemGetAllocationGranularity(&r->gran, &prop, hipMemAllocationGranularityRecommended));
    // Pieces meet at 2 MB boundaries, whatever the (4 KB) mapping granularity: the GPU's page tables describe contiguous virtual ranges by
    // fragments of up to 2 MB, and a shared piece that begins inside such a range was seen - once in five runs of the small test shapes,
    // never at 64 instances - to read through the neighbouring piece (a cell written through set 0 not visible through set 2).
    const size_t g = std::max<size_t>(r->gran, (size_t)2 << 20);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    auto create = [&](std::vector<h2e_ring::Piece>& v, size_t bytes) -> hipError_t {   // physical memory in pieces of at most 8 GB
        const size_t chunk = (size_t)8 << 30;
        for (size_t off = 0; off < bytes; off += chunk) {
            h2e_ring::Piece pc;
            pc.bytes = std::min(chunk, bytes - off);
            hipError_t e = hipMemCreate(&pc.h, pc.bytes, &prop, 0);
            if (e != hipSuccess) return e;
            v.push_back(pc);
        }
        return hipSuccess;
    };
    auto map = [&](char* va, const std::vector<h2e_ring::Piece>& v, size_t bytes) -> hipError_t {   // (HIP: a piece is mapped whole, offset 0)
        size_t done = 0;
        for (auto& pc : v) {
            hipError_t e = hipMemMap(va + done, pc.bytes, 0, pc.h, 0);
            if (e != hipSuccess) return e;
            done += pc.bytes;
        }
        return done == bytes ? hipSuccess : hipErrorInvalidValue;
    };
    const uint64_t rows[3] = {p->base_rows, p->range_rows, p->select_rows};
    const uint64_t row0[3] = {bs.base0, bs.range0, bs.select0};
    const uint64_t drow[3] = {bs.dbase, bs.drange, bs.dselect};
    const uint32_t cols[3] = {5, 3, 2};
    for (int a = 0; a < 3; a++) {
        h2e_ring::Arr& A = r->arr[a];
        const size_t rowb = (size_t)cols[a] * 32 * n_instances;
        A.row0 = row0[a];
        A.row1 = row0[a] + (uint64_t)bs.n_strands * drow[a];
        A.total = (std::max<size_t>(1, rows[a]) * rowb + g - 1) / g * g;
        A.a0 = (A.row0 * rowb + g - 1) / g * g;
        A.a1 = A.row1 * rowb / g * g;
        if (A.a1 <= A.a0) A.a0 = A.a1 = 0;   // (nothing worth sharing in this array)
        const size_t big_b = A.a1 - A.a0;
        if (big_b)
            for (int c = 0; c < 2; c++) HIP_TRY(create(A.big[c], big_b));
        A.rest.resize(depth);
        A.rest2.resize(depth);
        for (uint32_t c = 0; c < depth; c++) {
            HIP_TRY(create(A.rest[c], big_b ? A.a0 : A.total));
            if (big_b) HIP_TRY(create(A.rest2[c], A.total - A.a1));
        }
        A.va.assign(r->n_virtual, nullptr);
        for (uint32_t v = 0; v < r->n_virtual; v++) {
            void* va = nullptr;
            HIP_TRY(hipMemAddressReserve(&va, A.total, g, nullptr, 0));
            A.va[v] = va;
            const size_t first_b = big_b ? A.a0 : A.total;
            if (first_b) HIP_TRY(map((char*)va, A.rest[v % depth], first_b));
            if (big_b) HIP_TRY(map((char*)va + A.a0, A.big[v % 2], big_b));
            if (big_b && A.total > A.a1) HIP_TRY(map((char*)va + A.a1, A.rest2[v % depth], A.total - A.a1));
            HIP_TRY(hipMemSetAccess(va, A.total, &acc, 1));
        }
    }
    *out = r.release();
    return 0;
}

// out[0..2] = bytes of a full array set (base, range, select), [3..5] = bytes of each that the two shared copies back, [6] = physical bytes
// of the whole ring, [7] = the shared launch (index as h2e_program_launches lists them), [8] = virtual array sets, [9] = depth
int h2e_ring_info(const h2e_ring* r, uint64_t* out, uint32_t cap) {
    if (!r || !out) return fail(H2E_ERR_INVALID, "null argument");
    uint64_t v[10] = {0};
    for (int a = 0; a < 3; a++) {
        v[a] = r->arr[a].total;
        v[3 + a] = r->arr[a].a1 - r->arr[a].a0;
        v[6] += 2 * (r->arr[a].a1 - r->arr[a].a0) + (uint64_t)r->depth * (r->arr[a].total - (r->arr[a].a1 - r->arr[a].a0));
    }
    v[7] = (uint64_t)r->big_launch;
    v[8] = r->n_virtual;
    v[9] = r->depth;
    for (uint32_t i = 0; i < cap && i < 10; i++) out[i] = v[i];
    return 10;
}

int h2e_ring_arrays(const h2e_ring* r, uint64_t k, void** d_base, void** d_range, void** d_select) {
    if (!r || !d_base || !d_range || !d_select) return fail(H2E_ERR_INVALID, "null argument");
    const uint32_t v = (uint32_t)(k % r->n_virtual);
    *d_base = r->arr[0].va[v];
    *d_range = r->arr[1].va[v];
    *d_select = r->arr[2].va[v];
    return 0;
}

static int ring_submit(h2e_ring* r, uint64_t k, const void* d_inputs, void* d_status, void* d_digests, void* stream_, int* job) {
    if (!r || !job) return fail(H2E_ERR_INVALID, "null argument");
    *job = -1;
    if (k != r->next) return fail(H2E_ERR_INVALID, "h2e_ring_submit: runs of a ring are submitted in order (k = 0, 1, 2, ...)");
    h2e_ctx* ctx = r->ctx;
    RunFence fence;
    {
        std::lock_guard<std::mutex> guard(ctx->mu);
        if (ctx->depth != r->depth) return fail(H2E_ERR_INVALID, "h2e_ring_submit: the context's pipeline depth is no longer the ring's");
        // run k - 2 wrote the same physical rows of the shared launch: its completion event, as recorded for that run (the slot is
        // not re-used before run k - 2 + depth > k)
        if (k >= 2 && r->jobs[(k - 2) % r->depth] >= 0) {
            fence.seg = r->big_seg;
            // ... or, when the consumer of run k - 2 said where its reads end (h2e_ring_release), that point of ITS stream - which lies
            // behind the run's completion (the consumer waited for it with h2e_wait)
            const uint32_t q = (uint32_t)((k - 2) % r->depth);
            fence.ev = r->released_k[q] == k - 2 ? r->released[q] : ctx->slots[r->jobs[q]].done;
        }
    }
    // the other rows of set k were run k - depth's: if its consumer said where its reads end, this run starts behind that point
    // (otherwise the caller orders the re-use through the submission stream, as with h2e_submit)
    if (k >= r->depth && r->released_k[k % r->depth] == k - r->depth) {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipStreamWaitEvent((hipStream_t)stream_, r->released[k % r->depth], 0));
    }
    void *b, *g, *s;
    h2e_ring_arrays(r, k, &b, &g, &s);
    int rc = run_impl(ctx, r->prog, r->n_instances, d_inputs, b, g, s, d_status, (hipStream_t)stream_, false, job, d_digests, nullptr,
                      fence.ev ? &fence : nullptr);
    if (rc) return rc;
    r->jobs[k % r->depth] = *job;
    r->next = k + 1;
    return 0;
}
// The consumer of run k is done with the run's arrays at this point of `stream`: run k + 2, which writes the same physical rows of the
// shared launch, waits for it (instead of for run k's completion alone).  Call it before h2e_ring_submit(k + 2).
int h2e_ring_release(h2e_ring* r, uint64_t k, void* stream) {
    if (!r) return fail(H2E_ERR_INVALID, "null ring");
    if (k >= r->next) return fail(H2E_ERR_INVALID, "h2e_ring_release: run k has not been submitted");
    if (k + 2 < r->next) return fail(H2E_ERR_INVALID, "h2e_ring_release: run k + 2 has been submitted already");
    HIP_TRY(hipSetDevice(r->ctx->device));
    const uint32_t q = (uint32_t)(k % r->depth);
    if (!r->released[q]) HIP_TRY(hipEventCreateWithFlags(&r->released[q], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(r->released[q], (hipStream_t)stream));
    r->released_k[q] = k;
    return 0;
}
int h2e_ring_submit(h2e_ring* r, uint64_t k, const void* d_inputs, void* d_status, void* stream, int* job) {
    return ring_submit(r, k, d_inputs, d_status, nullptr, stream, job);
}
int h2e_ring_submit_digest(h2e_ring* r, uint64_t k, const void* d_inputs, void* d_status, void* d_digests, void* stream, int* job) {
    if (!d_digests) return fail(H2E_ERR_INVALID, "d_digests is null");
    return ring_submit(r, k, d_inputs, d_status, d_digests, stream, job);
}
