// Running a program: the packed expansion's order tables, the program's device copy, streams, and run_impl - the schedule of a run's
// value chain, expansions and fix-ups over the context's streams - with h2e_run / h2e_submit / h2e_wait.
// Included by h2e_capi.cpp INSIDE its extern "C" block.
// Order tables of the packed expansion (tape.h H2ELaunch::pk_order).  A wave of h2e_run_tape_packed takes G sub-ranges and every
// step runs ONE opcode for the groups whose cursor shows it, so a wave of G different opcode sequences costs up to the sum of
// them.  The programs it serves repeat themselves (a pairing check: 8 649 sub-ranges, 391 different opcode sequences), so the
// sub-ranges are classed by their sequence, every wave takes sub-ranges of one class (a class's last wave is padded with empty
// slots), and the heaviest waves are dispatched first.  Replayed on the real tapes (exp/pack_sim.py): the longest SIMD's work
// falls 2.0-2.9 x against taking the sub-ranges in tape order.
static void pack_orders_of(const h2e::Recorder& r, const h2e::Segment& sg, const uint32_t* subs, uint32_t n_sub, std::vector<uint32_t>& out,
                           std::array<uint32_t, 5>& off, std::array<uint32_t, 5>& n_waves) {
    // what an op costs the wave ~ the cells it writes
    auto op_cost = [](uint16_t opc) -> uint32_t {
        switch (opc) {
            case H2E_OP_DIV_CORE: return 140;
            case H2E_OP_INT_MUL: return 125;
            case H2E_OP_REDUCE: case H2E_OP_IS_INT_ZERO: return 40;
            case H2E_OP_ASSIGN_W: case H2E_OP_DECOMPOSE_NATIVE: return 23;
            case H2E_OP_BISEC_INT: case H2E_OP_SELECT_POINT: return 20;
            case H2E_OP_INT_ADD: case H2E_OP_INT_SUB: case H2E_OP_MASK_INT: return 13;
            case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: case H2E_OP_CACHE_INT: return 10;
            default: return 4;
        }
    };
    struct Class { uint64_t cost; std::vector<uint32_t> members; };
    std::vector<Class> classes;
    std::unordered_map<uint64_t, std::vector<uint32_t>> by_hash;   // hash -> classes with it (compared op by op: a collision must not mix sequences)
    auto same_sequence = [&](uint32_t a, uint32_t b) {
        if (subs[a + 1] - subs[a] != subs[b + 1] - subs[b]) return false;
        for (uint32_t i = 0; i < subs[a + 1] - subs[a]; i++)
            if (r.tape[sg.tape_begin + subs[a] + i].opcode != r.tape[sg.tape_begin + subs[b] + i].opcode) return false;
        return true;
    };
    for (uint32_t k = 0; k < n_sub; k++) {
        uint64_t h = 0xcbf29ce484222325ull, cost = 0;
        for (uint32_t o = subs[k]; o < subs[k + 1]; o++) {
            uint16_t opc = r.tape[sg.tape_begin + o].opcode;
            h = (h ^ opc) * 0x100000001b3ull;
            cost += op_cost(opc);
        }
        std::vector<uint32_t>& cand = by_hash[h];
        uint32_t cls = ~0u;
        for (uint32_t c : cand)
            if (same_sequence(classes[c].members[0], k)) cls = c;
        if (cls == ~0u) {
            cls = (uint32_t)classes.size();
            cand.push_back(cls);
            classes.push_back({cost, {}});
        }
        classes[cls].members.push_back(k);
    }
    std::vector<uint32_t> by_cost(classes.size());
    for (uint32_t c = 0; c < classes.size(); c++) by_cost[c] = c;
    std::stable_sort(by_cost.begin(), by_cost.end(), [&](uint32_t a, uint32_t b) { return classes[a].cost > classes[b].cost; });
    for (int k = 0; k < 5; k++) {
        const uint32_t G = 2u << k;
        off[k] = (uint32_t)out.size();
        uint32_t waves = 0;
        for (uint32_t c : by_cost) {
            const std::vector<uint32_t>& m = classes[c].members;
            for (size_t i = 0; i < m.size(); i += G, waves++)
                for (uint32_t g = 0; g < G; g++) out.push_back(i + g < m.size() ? m[i + g] : ~0u);
        }
        n_waves[k] = waves;
    }
}
static void build_pack_orders(h2e_program* p) {
    const h2e::Recorder& r = *p->rec;
    p->h_pk_order.clear();
    p->seg_pk_off.assign(r.segments.size(), std::array<uint32_t, 5>{0, 0, 0, 0, 0});
    p->seg_pk_waves.assign(r.segments.size(), std::array<uint32_t, 5>{0, 0, 0, 0, 0});
    for (size_t si = 0; si < r.segments.size(); si++) {
        const h2e::Segment& sg = r.segments[si];
        uint32_t n_sub = p->seg_n_sub[si];
        if (n_sub < 2 || sg.n_strands > 32 || n_sub > (1u << 18)) continue;   // (a packed launch has at most 32 lanes per sub-range)
        pack_orders_of(r, sg, p->h_subs.data() + p->seg_sub_begin[si], n_sub, p->h_pk_order, p->seg_pk_off[si], p->seg_pk_waves[si]);
    }
}

// The order tables only matter to launches with n_strands x n_instances <= 32 lanes per sub-range (engine.hip: the packed
// expansion).  BASELINE-sized batches, the MSM's segments and most of an operator-API context's cached op programs never take
// that path, so the tables (an op-by-op classification of the tape, five tables per segment, a device allocation) are made
// by the first run that does.
static int ensure_pack_orders(h2e_program* p, uint32_t n_instances) {
    if (p->pk_built) return 0;
    const h2e::Recorder& r = *p->rec;
    bool need = false;
    for (size_t si = 0; si < r.segments.size() && si < p->seg_n_sub.size(); si++)
        need = need || (p->seg_n_sub[si] >= 2 && (uint64_t)r.segments[si].n_strands * n_instances <= 32);
    if (!need) return 0;
    build_pack_orders(p);
    if (!p->h_pk_order.empty()) {
        HIP_TRY(hipMalloc((void**)&p->d_pk_order, p->h_pk_order.size() * 4));
        HIP_TRY(hipMemcpy(p->d_pk_order, p->h_pk_order.data(), p->h_pk_order.size() * 4, hipMemcpyHostToDevice));
    }
    p->pk_built = true;
    return 0;
}

static int ensure_device_program(h2e_ctx* ctx, h2e_program* p) {
    if (p->device == ctx->device) return 0;
    if (p->device >= 0) return fail(H2E_ERR_INVALID, "program already bound to another device");
    h2e::Recorder& r = *p->rec;
    HIP_TRY(hipSetDevice(ctx->device));
    auto up = [&](void** d, const void* h, size_t bytes) -> hipError_t {
        if (bytes == 0) bytes = 16;
        hipError_t e = hipMalloc(d, bytes);
        if (e != hipSuccess) return e;
        if (h) return hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
        return hipSuccess;
    };
    HIP_TRY(up((void**)&p->d_tape, r.tape.empty() ? nullptr : r.tape.data(), r.tape.size() * sizeof(H2EOp)));
    HIP_TRY(up((void**)&p->d_aux, r.aux.empty() ? nullptr : r.aux.data(), r.aux.size() * 4));
    HIP_TRY(up((void**)&p->d_pool, r.pool.empty() ? nullptr : r.pool.data(), r.pool.size() * 8));
    HIP_TRY(up((void**)&p->d_params, r.params.empty() ? nullptr : r.params.data(), r.params.size() * 4));
    HIP_TRY(up((void**)&p->d_fixups, r.fixups.empty() ? nullptr : r.fixups.data(), r.fixups.size() * 4));
    HIP_TRY(up((void**)&p->d_pre_args, r.pre_args.empty() ? nullptr : r.pre_args.data(), r.pre_args.size() * 4));
    p->h_subs.clear();
    p->seg_sub_begin.assign(r.segments.size(), ~0u);
    p->seg_n_sub.assign(r.segments.size(), 0);
    for (size_t si = 0; si < r.segments.size(); si++) {
        const h2e::Segment& sg = r.segments[si];
        uint32_t n_ops = sg.tape_end - sg.tape_begin;
        if (sg.n_cuts == 0 || n_ops == 0) continue;
        p->seg_sub_begin[si] = (uint32_t)p->h_subs.size();
        p->h_subs.push_back(0);
        uint32_t n = 0;
        for (uint32_t k = 0; k < sg.n_cuts; k++) {
            uint32_t at = r.cuts[sg.cuts_begin + k];
            if (at > p->h_subs.back() && at < n_ops) {
                p->h_subs.push_back(at);
                n++;
            }
        }
        p->h_subs.push_back(n_ops);
        p->seg_n_sub[si] = n + 1;
    }
    HIP_TRY(up((void**)&p->d_subs, p->h_subs.empty() ? nullptr : p->h_subs.data(), p->h_subs.size() * 4));
    // (the packed expansion's order tables are built by the first run that takes the packed path: ensure_pack_orders)
    HIP_TRY(up((void**)&p->d_vtape, p->h_vtape.empty() ? nullptr : p->h_vtape.data(), p->h_vtape.size() * sizeof(H2EVRec)));
    HIP_TRY(up((void**)&p->d_vpieces, p->h_vpieces.empty() ? nullptr : p->h_vpieces.data(), p->h_vpieces.size() * 4));
    HIP_TRY(up((void**)&p->d_lrecs, p->h_lrecs.empty() ? nullptr : p->h_lrecs.data(), p->h_lrecs.size() * sizeof(H2EVRec)));
    HIP_TRY(up((void**)&p->d_lrefs, p->h_lrefs.empty() ? nullptr : p->h_lrefs.data(), p->h_lrefs.size() * 4));
    HIP_TRY(up((void**)&p->d_lrounds, p->h_lrounds.empty() ? nullptr : p->h_lrounds.data(), p->h_lrounds.size() * 4));
    HIP_TRY(up((void**)&p->d_swords, p->h_swords.empty() ? nullptr : p->h_swords.data(), p->h_swords.size() * 4));
    HIP_TRY(up((void**)&p->d_soffsets, p->h_soffsets.empty() ? nullptr : p->h_soffsets.data(), p->h_soffsets.size() * 4));
    HIP_TRY(up((void**)&p->d_sktab, p->h_sktab.empty() ? nullptr : p->h_sktab.data(), p->h_sktab.size() * 8));
    HIP_TRY(up((void**)&p->d_sext, p->h_sext.empty() ? nullptr : p->h_sext.data(), p->h_sext.size() * 4));
    p->device = ctx->device;
    return 0;
}

// One run.  `join` = true: the caller's stream completes when every stream of the run has (h2e_run); false: the
// caller's stream only carries the value chain and `slot.done` is recorded on the fix-up stream when the run is
// complete (h2e_submit / h2e_wait).
// kind: 0 expansion, 1 value chain / side, 2 fix-up
// (CU masks for the value-chain streams - H2E_CU_RESERVE, rounds 3 and 4 - were a measured loser and are gone: a CU-masked expansion
// stream is slow in itself, 21.5 ms per MSM step with 8 CUs set aside, and masked streams of the pairing batches did not overlap at all)
static hipError_t make_stream(h2e_ctx* ctx, hipStream_t* out, int prio, int kind) {
    (void)ctx;
    (void)kind;
    return hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio);
}

// A run made of several caller batches (h2e_submit_batches): k batches of arr_n instances each, every batch with its own inputs,
// arrays and status words; the run's instance i is instance i % arr_n of batch i / arr_n.
struct RunBatches {
    uint32_t k = 0, arr_n = 0;
    const void* const* inputs = nullptr;
    void* const* base = nullptr;
    void* const* range = nullptr;
    void* const* select = nullptr;
    void* const* status = nullptr;
};
// One more edge in a run's schedule (ring.hpp): whatever of the run writes the rows of segment `seg` - its value chain's stores, its
// expansion, its fix-ups, all ordered behind the chain stream at that point - waits for `ev` first.
struct RunFence {
    int seg = -1;
    hipEvent_t ev = nullptr;
};
// Column emission (h2e_run_columns): the run's expansions store halo2's per-instance advice columns themselves; the batch-interleaved
// arrays stay the working copy operands are read from.
struct RunColumns {
    void* col[3] = {nullptr, nullptr, nullptr};   // [instance][col][rows][4 words] per region
    int form = 0;
};
static int run_impl(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
                    void* d_select, void* d_status, hipStream_t stream, bool join, int* slot_out, void* d_digests = nullptr,
                    const RunBatches* batches = nullptr, const RunFence* fence = nullptr, const RunColumns* columns = nullptr) {
    if (!ctx || !p) return fail(H2E_ERR_INVALID, "null ctx/program");
    if (!batches && (!d_inputs || !d_base || !d_range || !d_select || !d_status)) return fail(H2E_ERR_INVALID, "null device pointer");
    std::lock_guard<std::mutex> guard(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_device_program(ctx, p);
    if (rc) return rc;
    if ((rc = ensure_pack_orders(p, n_instances))) return rc;
    int fp = p->field_pair;
    h2e::Recorder& r = *p->rec;
    {   // constants of every W field the program's segments work in (a GeneralScalarEccContext has two)
        bool need[3] = {false, false, false};
        need[fp] = true;
        for (auto& sg : r.segments) need[sg.field_pair] = true;
        for (int f = 0; f < 3; f++)
            if (need[f] && !ctx->d_fc[f]) {
                HIP_TRY(hipMalloc((void**)&ctx->d_fc[f], sizeof(H2EFieldConsts)));
                HIP_TRY(hipMemcpy(ctx->d_fc[f], &field_pair(f).fc, sizeof(H2EFieldConsts), hipMemcpyHostToDevice));
                HIP_TRY((hipError_t)h2e_engine_set_consts(f, &field_pair(f).fc));
            }
    }
    if (columns) {
        if (n_instances % 64 != 0) return fail(H2E_ERR_INVALID, "h2e_run_columns: n_instances must be a multiple of 64 (a wave is 64 instances at one row)");
        if (!ctx->cols_consts) {   // (every column unit has its own constant memory: each gets the pairs its segments may work in)
            for (int f = 0; f < 3; f++) {
                HIP_TRY((hipError_t)h2e_engine_set_consts_colsfp0(f, &field_pair(f).fc));
                HIP_TRY((hipError_t)h2e_engine_set_consts_colsfp1(f, &field_pair(f).fc));
                HIP_TRY((hipError_t)h2e_engine_set_consts_colsfp2(f, &field_pair(f).fc));
            }
            ctx->cols_consts = true;
        }
    }
    if (!ctx->expand_stream) HIP_TRY(make_stream(ctx, &ctx->expand_stream, ctx->prio_expand, 0));
    if (!ctx->fixup_stream) HIP_TRY(make_stream(ctx, &ctx->fixup_stream, ctx->prio_fixup, 2));
    // Streams.  The *value chain* (predictor kernels + values-only replay, or the plain tape for segments without cuts)
    // runs on the caller's stream: it is what later segments depend on.  The full expansion of a cut segment only needs
    // the value chain up to that segment, so it runs on a second stream and overlaps the value chain of the following
    // segments (and, with h2e_submit, of the following run); predictors that only depend on earlier predictors, fork
    // segments outside the chain and the program's serial tail on the slot's side stream; inverse fix-ups on a fourth.
    // (The runtime maps a process's streams onto GPU_MAX_HW_QUEUES = 4 hardware queues by default; streams that share a
    // queue serialise.  A host that pipelines runs should raise it to 8 before HIP initialises - bench.py does.)
    int slot_index = (int)(ctx->n_runs % ctx->depth);
    JobSlot& J = ctx->slots[slot_index];
    {   // the side stream only exists for programs that use it (every stream takes one of the process's hardware queues)
        bool need_side = p->tail_from >= 0;
        for (auto& pk : r.pre_kernels) need_side = need_side || pk.early_after_segment >= 0;
        for (size_t si = 0; si < r.segments.size() && si < p->seg_side_dep.size(); si++) need_side = need_side || p->seg_side_dep[si] != -2;
        if (need_side && !J.side_stream) HIP_TRY(make_stream(ctx, &J.side_stream, ctx->prio_side, 1));
    }
    if (!join && !J.chain_stream) HIP_TRY(make_stream(ctx, &J.chain_stream, ctx->prio_side, 1));
    if (!join) {   // the chain stream takes over from the caller's stream at this point
        if (!J.order_ev) HIP_TRY(hipEventCreateWithFlags(&J.order_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(J.order_ev, stream));
        HIP_TRY(hipStreamWaitEvent(J.chain_stream, J.order_ev, 0));
    }
    const hipStream_t sa_main = join ? stream : J.chain_stream;
    hipStream_t sa = sa_main, sb = ctx->expand_stream, sc = J.side_stream, sd = ctx->fixup_stream;
    {
        // Pipelined runs whose expansions are big but not huge (a full 64-check bn256 batch: two launches of ~5 500 waves, 1.2-1.7 ms each)
        // alternate between TWO expansion streams by slot: on one stream ~0.1 ms of event latency stood between any two launches, and the
        // stream was the step (device timeline: busy 3.0 of every 3.3 ms) - 64 x bn256 3.07 -> 2.93 ms per step.  An expansion that is split
        // into parts (the MSM windows: 10 ms of stores) keeps the one stream: two of those side by side only share the store bandwidth
        // (15.48 -> 15.6 ms).  H2E_SCHED & 1024: one stream for all (A/B)
        bool huge_x = false;
        for (size_t si = 0; si < r.segments.size() && si < p->seg_n_sub.size(); si++)
            huge_x = huge_x || (p->seg_n_sub[si] > 1 && (uint64_t)p->seg_n_sub[si] * r.segments[si].n_strands * n_instances >= ctx->x_split_min_lanes);
        if (!join && !huge_x && !(ctx->sched & 1024u) && (slot_index & 1)) {
            if (!ctx->expand_stream2) HIP_TRY(make_stream(ctx, &ctx->expand_stream2, ctx->prio_expand, 0));
            sb = ctx->expand_stream2;
        }
    }
#ifdef H2E_DEBUG_HOOKS
    if (FILE* f = dbg_log_file()) {
        fprintf(f, "run %llu begin slot %d join %d streams chain %p expand %p side %p fixup %p small %p\n", (unsigned long long)ctx->n_runs + 1, slot_index, join ? 1 : 0,
                (void*)sa, (void*)sb, (void*)sc, (void*)sd, (void*)ctx->small_stream);
        fflush(f);
    }
#endif
    ctx->n_runs++;
#ifdef H2E_DEBUG_HOOKS
    g_dbg_run = ctx->n_runs;
#endif
    ctx->last_slot = slot_index;
    if (slot_out) *slot_out = slot_index;
    if (!J.done) HIP_TRY(hipEventCreateWithFlags(&J.done, hipEventDisableTiming));
    // the slot's previous run (two submissions ago) must be complete before its workspace is overwritten
    if (J.used) HIP_TRY(hipStreamWaitEvent(sa, J.done, 0));
    J.used = true;
    // instance descriptors
    if (J.inst_cap < n_instances) {
        if (J.d_inst) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipFree(J.d_inst));
            J.d_inst = nullptr;
        }
        HIP_TRY(hipMalloc((void**)&J.d_inst, (size_t)n_instances * sizeof(InstanceDescHost)));
        J.inst_cap = n_instances;
    }
    size_t slot_words = r.fp.w_words;
    // workspace
    auto grow = [&](uint64_t** buf, size_t* have, size_t need) -> hipError_t {
        if (need <= *have) return hipSuccess;
        if (*buf) {
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) return e;
            (void)hipFree(*buf);
            *buf = nullptr;
        }
        hipError_t e = hipMalloc((void**)buf, need * 8);
        if (e == hipSuccess) *have = need;
        return e;
    };
    // the value chain's workspace is instance-minor ([slot][instance][w words], engine.hip InstanceDesc); a slot holds a
    // value of the widest W field the program works in
    size_t wsw = slot_words;
    for (auto& sg : r.segments) wsw = std::max<size_t>(wsw, (size_t)field_pair(sg.field_pair).w_words);
    size_t hint_words = ((size_t)r.n_hint_slots + H2E_ECC_HINT_SLOTS) * wsw,  // spare: the replay prefetches slot + 8
           nd_words = hint_words * 2,
           jac_words = (size_t)r.n_jac_slots * 3 * wsw,
           sel_words = (size_t)r.n_sel_slots * H2E_SEL_SLOTS * wsw;
    HIP_TRY(grow(&J.ws_hints, &J.ws_hints_words, std::max<size_t>(1, hint_words * n_instances)));
    HIP_TRY(grow(&J.ws_nd, &J.ws_nd_words, std::max<size_t>(1, nd_words * n_instances)));
    HIP_TRY(grow(&J.ws_jac, &J.ws_jac_words, std::max<size_t>(1, jac_words * n_instances)));
    HIP_TRY(grow(&J.ws_sel, &J.ws_sel_words, std::max<size_t>(1, sel_words * n_instances)));
    {   // the table of per-instance descriptors, written on the device (handoff.hip: every field is affine in the instance index; no
        // pinned staging table, so nothing here makes the host wait for an earlier run of the slot)
        // batch-interleaved advice arrays [row][col][half][instance][2 words]: instance i starts 2 words in
        // (several caller batches in one run: the caller-side pointers per batch, inside a batch affine as before)
        const uint64_t first[9] = {0, 0, 0, 0, 0, (uint64_t)J.ws_hints, (uint64_t)J.ws_nd, (uint64_t)J.ws_jac, (uint64_t)J.ws_sel};
        const uint64_t stride[9] = {16, 16, 16, (uint64_t)r.n_input_slots * slot_words * 8, 4, wsw * 8, wsw * 8, wsw * 8, wsw * 8};
        const uint32_t nb = batches ? batches->k : 1, arr_n = batches ? batches->arr_n : n_instances;
        std::vector<uint64_t> bfirst(5 * (size_t)nb);
        for (uint32_t b = 0; b < nb; b++) {
            bfirst[0 * nb + b] = (uint64_t)(batches ? batches->base[b] : d_base);
            bfirst[1 * nb + b] = (uint64_t)(batches ? batches->range[b] : d_range);
            bfirst[2 * nb + b] = (uint64_t)(batches ? batches->select[b] : d_select);
            bfirst[3 * nb + b] = (uint64_t)(batches ? batches->inputs[b] : d_inputs);
            bfirst[4 * nb + b] = (uint64_t)(batches ? batches->status[b] : d_status);
        }
        int trc = h2e_engine_instance_table(J.d_inst, n_instances, arr_n, bfirst.data(), first, stride, (uint32_t)(n_instances * wsw), sa);
        if (trc < 0) return fail(H2E_ERR_INVALID, "instance table: bad batch geometry");
        if (trc != 0) return fail(H2E_ERR_HIP, std::string("instance table kernel launch failed: ") + hipGetErrorString((hipError_t)trc));
    }
#ifdef H2E_DEBUG_HOOKS
    dbg_mark("table queued");
#endif
    DBG_STAMP(1001u, sa);
    // stream digest: the expansion and fix-up kernels of this run add to it (every other stream starts behind this point)
    if (d_digests) {
        if (J.dg_cap < n_instances) {
            if (J.dg_shards) {
                HIP_TRY(hipDeviceSynchronize());
                HIP_TRY(hipFree(J.dg_shards));
                J.dg_shards = nullptr;
            }
            HIP_TRY(hipMalloc((void**)&J.dg_shards, (size_t)H2E_DG_SHARDS * 3 * n_instances * 4 * sizeof(uint64_t)));
            J.dg_cap = n_instances;
        }
        HIP_TRY(hipMemsetAsync(J.dg_shards, 0, (size_t)H2E_DG_SHARDS * 3 * n_instances * 4 * sizeof(uint64_t), sa));
    }
    bool used_sd = false;
    std::vector<hipEvent_t> early_done(r.pre_kernels.size(), nullptr);
    size_t n_sync = 0;
    auto sync_event = [&]() -> hipEvent_t {
        if (n_sync == J.sync_ev.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            J.sync_ev.push_back(e);
        }
        return J.sync_ev[n_sync++];
    };
    auto prof_event = [&](uint32_t k) -> hipEvent_t {
        while (J.ev.size() <= k) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            J.ev.push_back(e);
        }
        return J.ev[k];
    };
    const bool profiling = ctx->profiling;
    J.profiled = profiling;
    {   // the engine's streams start after everything already queued on the caller's stream
        hipEvent_t e = sync_event();
        HIP_TRY(hipEventRecord(e, sa));
        HIP_TRY(hipStreamWaitEvent(sb, e, 0));
    }
    J.n_launches = 0;
    J.x_kernels.clear();
    ctx->last_split_segments = 0;
    hipStream_t se = sc;
    std::vector<hipEvent_t> seg_ev(r.segments.size(), nullptr), side_done(r.segments.size(), nullptr);
    hipEvent_t run_begin = sync_event();
    HIP_TRY(hipEventRecord(run_begin, sa));
    bool used_se = false, used_small = false, used_x = false, merged_x = false;
    H2ELaunch pending_L;
    uint32_t pending_li = 0;
    bool have_pending = false;
    // a held-back expansion goes out once the next long predictor chain (a segment with pre-selected points: the MSM
    // windows) is queued - next to that chain it costs 1 ms less than next to the select kernel before it - or, if there
    // is no such segment, right behind the next value chain
    bool hold_longer = false;
    auto flush_pending = [&]() -> int {   // launch an expansion that was held back behind a later value chain
        hipStream_t sp = sb;
        if (!join && (ctx->sched & 8u)) {   // pipelined: beside the previous run's big expansions, not between them
            if (!ctx->small_stream) HIP_TRY(make_stream(ctx, &ctx->small_stream, ctx->prio_expand, 0));
            sp = ctx->small_stream;
            used_small = true;
        }
        hipEvent_t e0 = sync_event();
        HIP_TRY(hipEventRecord(e0, sa));
        HIP_TRY(hipStreamWaitEvent(sp, e0, 0));
        if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * pending_li + 2), sp)); DBG_STAMP(4 * pending_li + 2, sp);
        int prc2 = columns ? h2e_engine_launch_cols(&pending_L, J.d_inst, n_instances, sp)
                           : H2E_LAUNCH((int)pending_L.field_pair, 2, &pending_L, J.d_inst, n_instances, ctx->d_fc[pending_L.field_pair], sp);
        if (prc2 != 0) return fail(H2E_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString((hipError_t)prc2));
        if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * pending_li + 3), sp)); DBG_STAMP(4 * pending_li + 3, sp);
        if (pending_L.n_fixups) {
            hipEvent_t e1 = sync_event();
            HIP_TRY(hipEventRecord(e1, sp));
            HIP_TRY(hipStreamWaitEvent(sd, e1, 0));
            used_sd = true;
            prc2 = H2E_LAUNCH((int)pending_L.field_pair, 4, &pending_L, J.d_inst, n_instances, ctx->d_fc[pending_L.field_pair], sd);
            if (prc2 != 0) return fail(H2E_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString((hipError_t)prc2));
        }
        have_pending = false;
        return 0;
    };
    // (the field chain reads the constant pool where the MSM predictors read strand parameters)
    auto pk_params = [&](const h2e::PreKernel& pk) -> const uint32_t* {
        return pk.k.kind == H2E_PRE_FIELD_CHAIN ? (const uint32_t*)p->d_pool : p->d_params;
    };
    // the digit chain in front of the next launched segment, if it has one (-1: none): it becomes ready together with an expansion
    // launched now, which then gets a gate (engine.hip h2e_gate; H2E_SCHED bit 32 switches the gates off, A/B)
    auto next_digit_chain = [&](size_t si) -> int {
        // (a pipelined run's expansions queue on streams they share with other runs' - a gate there would hold those up - and its
        // chains start next to other runs' expansions whatever it does: h2e_run only)
        if ((ctx->sched & 32u) || !join) return -1;
        size_t sj = si + 1;
        while (sj < r.segments.size() && r.segments[sj].tape_end <= r.segments[sj].tape_begin) sj++;
        if (sj >= r.segments.size()) return -1;
        for (size_t pi = 0; pi < r.pre_kernels.size(); pi++) {
            const h2e::PreKernel& pk = r.pre_kernels[pi];
            if (pk.before_segment == sj && pk.k.kind == H2E_PRE_FIELD_CHAIN && pk.k.f_mode == 1 && pk.early_after_segment < 0) return (int)pi;
        }
        return -1;
    };
    int gate_for = -1;   // the pre-kernel a gate launched in this run is waiting for
    // a gate counts on a chain that is launched later in this function: on EVERY exit on which that chain was not launched (an
    // error return in between, or no such chain) the slot's target goes back, or every later gate of the slot would spin its
    // whole timeout (the gate itself just times out)
    struct GateGuard {
        JobSlot& J;
        const int& gate_for;
        uint32_t n;
        ~GateGuard() { if (gate_for >= 0) J.gate_total -= n; }
    } gate_guard{J, gate_for, n_instances};
    bool run_has_big_x = false;
    for (size_t si = 0; si < r.segments.size(); si++)
        run_has_big_x = run_has_big_x || (p->seg_n_sub[si] > 1 && (uint64_t)p->seg_n_sub[si] * r.segments[si].n_strands * n_instances >= ctx->small_x_lanes);
    for (size_t si = 0; si < r.segments.size(); si++) {
        const h2e::Segment& s = r.segments[si];
        if (s.tape_end <= s.tape_begin) continue;
#ifdef H2E_DEBUG_HOOKS
        g_dbg_run = ctx->n_runs;
        g_dbg_si = (int)si;
#endif
        if ((int64_t)si == p->tail_from && sa == sa_main) {
            // the serial tail of the program: from here on the value chain continues on the slot's side stream, and the
            // caller's stream is free for the next run's value chain
            hipEvent_t e = sync_event();
            HIP_TRY(hipEventRecord(e, sa));
            HIP_TRY(hipStreamWaitEvent(sc, e, 0));
            sa = sc;
            used_se = true;
        }
        // side segments this one reads must be done
        for (size_t sj = 0; sj < si; sj++)
            if (side_done[sj] && p->seg_first_reader[sj] <= si) {
                HIP_TRY(hipStreamWaitEvent(sa, side_done[sj], 0));
                side_done[sj] = nullptr;
            }
        uint32_t li = J.n_launches;
        if (J.x_kernels.size() <= li) J.x_kernels.resize(li + 1, 1);
        if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 0), sa)); DBG_STAMP(4 * li + 0, sa);
        // this segment's predictors: chains first, then (after the early starters below) their finalize kernels
        for (size_t pi = 0; pi < r.pre_kernels.size(); pi++) {
            const h2e::PreKernel& pk = r.pre_kernels[pi];
            if (pk.before_segment != si) continue;
            if (early_done[pi]) {  // already running on the side stream: just wait for it
                HIP_TRY(hipStreamWaitEvent(sa, early_done[pi], 0));
                continue;
            }
            H2EPreKernel k1 = pk.k;
            if ((int)pi == gate_for) {   // a gate is waiting for this chain's workgroups
                k1.f_started = J.d_gate;
                gate_for = -1;
            }
            int prc = H2E_PREDICT(fp, 1, &k1, p->d_pre_args, pk_params(pk), p->d_aux, J.d_inst, n_instances, ctx->d_fc[fp], sa);
            if (prc != 0) return fail(H2E_ERR_HIP, std::string("predictor launch failed: ") + hipGetErrorString((hipError_t)prc));
        }
        if (have_pending && hold_longer && s.sel_stride) {
            int frc = flush_pending();
            if (frc) return frc;
        }
        // predictors of later segments that only depend on this segment's predictor chains start now, on the side stream
        for (size_t pi = 0; pi < r.pre_kernels.size(); pi++) {
            const h2e::PreKernel& pk = r.pre_kernels[pi];
            if (pk.early_after_segment != (int32_t)si || pk.before_segment <= si) continue;
            hipEvent_t e0 = sync_event();
            HIP_TRY(hipEventRecord(e0, sa));
            HIP_TRY(hipStreamWaitEvent(sc, e0, 0));
            int prc = H2E_PREDICT(fp, 3, &pk.k, p->d_pre_args, pk_params(pk), p->d_aux, J.d_inst, n_instances, ctx->d_fc[fp], sc);
            if (prc != 0) return fail(H2E_ERR_HIP, std::string("predictor launch failed: ") + hipGetErrorString((hipError_t)prc));
            hipEvent_t e1 = sync_event();
            HIP_TRY(hipEventRecord(e1, sc));
            early_done[pi] = e1;
            used_se = true;
        }
        for (size_t pi = 0; pi < r.pre_kernels.size(); pi++) {
            const h2e::PreKernel& pk = r.pre_kernels[pi];
            if (pk.before_segment != si || early_done[pi]) continue;
            int prc = H2E_PREDICT(fp, 2, &pk.k, p->d_pre_args, pk_params(pk), p->d_aux, J.d_inst, n_instances, ctx->d_fc[fp], sa);
            if (prc != 0) return fail(H2E_ERR_HIP, std::string("predictor launch failed: ") + hipGetErrorString((hipError_t)prc));
        }
        H2ELaunch L;
        L.tape = p->d_tape + s.tape_begin;
        L.n_ops = s.tape_end - s.tape_begin;
        L.n_strands = s.n_strands;
        L.strand_base0 = s.base0;
        L.strand_range0 = s.range0;
        L.strand_select0 = s.select0;
        L.delta_base = s.dbase;
        L.delta_range = s.drange;
        L.delta_select = s.dselect;
        L.input_stride = s.input_stride;
        L.n_params = s.n_params;
        L.params = p->d_params + s.params_begin;
        L.aux = p->d_aux;
        L.const_pool = p->d_pool;
        L.hint_stride = s.hint_stride;
        L.n_fixups = s.n_fixups;
        L.fixups = p->d_fixups + s.fixups_begin;
        L.rel_refs = s.is_fork ? 1 : 0;
        L.field_pair = (uint32_t)s.field_pair;
        L.slot_words = (uint32_t)slot_words;
        L.n_sub = p->seg_n_sub[si];
        L.sub = L.n_sub > 1 ? p->d_subs + p->seg_sub_begin[si] : nullptr;
        L.pk_order = nullptr;
        L.pk_n_sub = 0;
        if (L.n_sub > 1 && si < p->seg_pk_waves.size() && p->d_pk_order) {
            L.pk_order = p->d_pk_order;
            L.pk_n_sub = L.n_sub;
            for (int k = 0; k < 5; k++) {
                L.pk_off[k] = p->seg_pk_off[si][k];
                L.pk_waves[k] = p->seg_pk_waves[si][k];
            }
        }
        bool compiled = si < p->seg_n_pieces.size() && p->seg_n_pieces[si] > 0;
        L.vtape = compiled ? p->d_vtape : nullptr;
        L.vpieces = compiled ? p->d_vpieces + 2 * (size_t)p->seg_piece_begin[si] : nullptr;
        L.n_vpieces = compiled ? p->seg_n_pieces[si] : 0;
        L.v_int_slots = compiled ? p->seg_v_slots[si] : 0;
        L.v_units = compiled ? p->seg_v_units[si] : 0;
        L.sel_stride = s.sel_stride;
        bool levels = compiled && si < p->seg_l_steps.size() && p->seg_l_steps[si] > 0;
        L.lrecs = levels ? p->d_lrecs + p->seg_l_begin[si] : nullptr;
        L.lrefs = p->d_lrefs;
        L.lrounds = levels && p->seg_l_pair[si] == 2 ? p->d_lrounds + p->seg_lr_begin[si] : nullptr;
        L.l_recs = levels ? p->seg_l_recs[si] : 0;
        bool hstore = si < p->seg_n_sops.size() && p->seg_n_sops[si] > 0;
        L.s_words = hstore ? p->d_swords + p->seg_s_begin[si] : nullptr;
        L.s_offsets = hstore ? p->d_soffsets + p->seg_so_begin[si] : nullptr;
        L.s_ktab = hstore ? p->d_sktab + p->seg_sk_begin[si] : nullptr;
        L.n_sops = hstore ? p->seg_n_sops[si] : 0;
        L.s_ext = hstore && si < p->seg_sx_begin.size() ? p->d_sext + p->seg_sx_begin[si] : nullptr;
        L.dg_out = d_digests ? J.dg_shards : nullptr;
        L.dg_shards = H2E_DG_SHARDS;
        {
            const uint64_t rows3[3] = {p->base_rows, p->range_rows, p->select_rows};
            const uint64_t cols3[3] = {5, 3, 2};
            for (int q = 0; q < 3; q++) {
                L.col[q] = columns ? (uint64_t*)columns->col[q] : nullptr;
                L.col_stride[q] = cols3[q] * rows3[q] * 4;
                L.col_rows[q] = (uint32_t)rows3[q];
            }
            L.col_form = columns ? (uint32_t)columns->form : 0;
#ifdef H2E_DEBUG_HOOKS
            if (columns)   // timing experiments of the column unit (exp/r6_cols_dbg.sh: an engine built with exp/engine_experiments.patch reads bits 8..)
                if (const char* e = dbg_env("H2E_COLS_DBG")) L.col_form |= (uint32_t)atoi(e) << 8;
#endif
        }
        L.l_steps = levels ? p->seg_l_steps[si] : 0;
        L.l_slots = levels ? p->seg_l_slots[si] : 0;
        L.l_pair = levels ? p->seg_l_pair[si] : 0;
        int lrc;
        auto launch_one = [&](int mode, const H2ELaunch& l, hipStream_t st) -> int {
            // (the full expansion of a column-emission run: the column unit's kernel; value chains and fix-ups are the plain ones)
            int rc2 = (columns && mode == 2) ? h2e_engine_launch_cols(&l, J.d_inst, n_instances, st)
                                             : H2E_LAUNCH((int)l.field_pair, mode, &l, J.d_inst, n_instances, ctx->d_fc[l.field_pair], st);
            if (rc2 != 0) return fail(H2E_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString((hipError_t)rc2));
            return 0;
        };
        auto launch = [&](int mode, hipStream_t st) -> int { return launch_one(mode, L, st); };
        // A big expansion (the MSM windows) goes out as two back-to-back launches over the first x_split_pct percent / the
        // rest of its sub-ranges: while the first one drains, the value chain that became ready meanwhile (the MSM tail's
        // replay wants most of a CU's LDS per workgroup) gets its CUs instead of waiting for the whole expansion, and the
        // inverse fix-up of the first part runs under the second.
        // (x_parts > 2: the second part again in equal launches - the last inverse fix-up, which nothing can run under, shrinks with it)
        uint32_t split_sub = 0, split_fix = 0;
        std::vector<uint32_t> part_sub, part_fix;   // boundaries: sub-range index / fix-up index each launch starts at (+ the end)
        {
            uint32_t pct = ctx->x_split_pct;
            if (pct > 0 && pct < 100 && L.n_sub >= 4 && (uint64_t)L.n_sub * L.n_strands * n_instances >= ctx->x_split_min_lanes) {
                split_sub = std::min<uint32_t>(std::max<uint32_t>(2, (uint32_t)((uint64_t)L.n_sub * pct / 100)), L.n_sub - 2);
                auto fb = r.fixups.begin() + s.fixups_begin;
                const bool sorted_fix = std::is_sorted(fb, fb + s.n_fixups);
                // fix-up rows are recorded in tape order: those below the first row of a part belong to the parts before it
                auto fix_at = [&](uint32_t sub) -> uint32_t {
                    uint32_t row = r.tape[s.tape_begin + p->h_subs[p->seg_sub_begin[si] + sub]].base_row;
                    return sorted_fix ? (uint32_t)(std::lower_bound(fb, fb + s.n_fixups, row) - fb) : 0;
                };
                split_fix = fix_at(split_sub);
                part_sub = {0, split_sub};
                part_fix = {0, split_fix};
                uint32_t extra = std::min<uint32_t>(ctx->x_parts > 2 ? ctx->x_parts - 2 : 0, (L.n_sub - split_sub) / 2);
                for (uint32_t q = 1; q <= extra; q++) {
                    uint32_t at = split_sub + (uint32_t)((uint64_t)(L.n_sub - split_sub) * q / (extra + 1));
                    if (at > part_sub.back() && at < L.n_sub) {
                        part_sub.push_back(at);
                        part_fix.push_back(fix_at(at));
                    }
                }
                part_sub.push_back(L.n_sub);
                part_fix.push_back(s.n_fixups);
                ctx->last_split_segments++;
            }
        }
        // the inverse fix-up of a segment only touches cells nothing else reads or writes: own stream, after the expansion
        // (a small expansion keeps its fix-up in its own stream)
        // (the predictors above only touch the slot's workspace; from here on the segment's cells are written)
        if (fence && fence->ev && fence->seg == (int)si) HIP_TRY(hipStreamWaitEvent(sa, fence->ev, 0));
        bool fixup_in_stream = false;
        auto fixup_part = [&](hipStream_t st, uint32_t lo, uint32_t hi) -> int {
            if (hi <= lo) return 0;
            H2ELaunch f = L;
            f.fixups = L.fixups + lo;
            f.n_fixups = hi - lo;
            if (fixup_in_stream && !join && sc && (ctx->sched & 1u) && st != sc) {   // pipelined: keep the shared expansion stream free
                hipEvent_t e = sync_event();
                HIP_TRY(hipEventRecord(e, st));
                HIP_TRY(hipStreamWaitEvent(sc, e, 0));
                used_se = true;
                return launch_one(4, f, sc);
            }
            // A pipelined run WITHOUT a big expansion (a pairing batch smaller than half a wave) sends its small fix-ups to the fix-up
            // stream as well: its expansions share one stream with those of the other runs in flight, and a 0.1 ms one-workgroup
            // inversion behind every one of them made that stream the step (16 x bls12_381 at four runs in flight 1.81 -> 1.58 ms, 8 x
            // bn256 1.40 -> 1.10, 2 x bls12_381 1.22 -> 0.95; the MSM, whose small expansions run beside big ones: no difference, left
            // as it was).  H2E_SCHED & 64: in their stream as before (A/B)
            if (fixup_in_stream && (join || (run_has_big_x && !(ctx->sched & 128u)) || (ctx->sched & 64u) || (used_x && st == J.x_stream) || (merged_x && st == sa))) return launch_one(4, f, st);   // (128: to the fix-up stream whatever the run holds - experiment)
            hipEvent_t e = sync_event();
            HIP_TRY(hipEventRecord(e, st));
            HIP_TRY(hipStreamWaitEvent(sd, e, 0));
            used_sd = true;
            return launch_one(4, f, sd);
        };
        auto fixup_after = [&](hipStream_t st) -> int { return fixup_part(st, 0, s.n_fixups); };
        // full expansion + fix-up of this segment on stream st
        auto expand = [&](hipStream_t st) -> int {
            int xrc;
            if (!split_sub) {
                if ((xrc = launch(2, st))) return xrc;
                return 0;
            }
            const size_t n_parts = part_sub.size() - 1;
            J.x_kernels[li] = (uint32_t)n_parts;
            for (size_t q = 0; q < n_parts; q++) {
                H2ELaunch a = L;
                a.n_sub = part_sub[q + 1] - part_sub[q];
                a.sub = L.sub + part_sub[q];
                if ((xrc = launch_one(2, a, st))) return xrc;
                if (q + 1 < n_parts && (xrc = fixup_part(st, part_fix[q], part_fix[q + 1]))) return xrc;   // (the last part's: expand_fixup)
            }
            return 0;
        };
        auto expand_fixup = [&](hipStream_t st) -> int { return fixup_part(st, split_sub ? part_fix[part_fix.size() - 2] : 0, s.n_fixups); };
        if (L.n_sub > 1) {
            if ((lrc = launch(1, sa))) return lrc;
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 1), sa)); DBG_STAMP(4 * li + 1, sa);
            if (have_pending && !hold_longer && (lrc = flush_pending())) return lrc;
            // (H2E_SCHED & 16: pipelined runs do not hold the expansion back - measured 0.35 ms per step worse)
            if (s.expand_after_next && si + 1 < r.segments.size() && p->seg_n_sub[si + 1] > 1 && (join || !(ctx->sched & 16u))) {
                if (have_pending && (lrc = flush_pending())) return lrc;   // single slot: never overwrite a held expansion
                pending_L = L;
                pending_li = li;
                have_pending = true;
                hold_longer = false;
                for (size_t sj = si + 1; sj < r.segments.size(); sj++) hold_longer = hold_longer || r.segments[sj].sel_stride != 0;
                J.n_launches++;
                seg_ev[si] = sync_event();
                HIP_TRY(hipEventRecord(seg_ev[si], sa));
                continue;
            }
            // a small expansion (the MSM tail: 763 waves) queues behind the big one on the expansion stream: beside it on
            // the side stream it and its fix-up slow the big one down by more than they take alone; its fix-up follows it
            // in its stream: the fix-up stream still holds the big expansion's second fix-up
            bool small_x = (uint64_t)L.n_sub * L.n_strands * n_instances < ctx->small_x_lanes;
            hipStream_t sx = (small_x && !join && sc && (ctx->sched & 2u)) ? sc : sb;
            if (sx == sc) used_se = true;
            if (small_x && !join && (ctx->sched & 4u)) {
                // pipelined: the shared expansion stream only carries the big expansions - the small ones (latency-bound: a few
                // hundred waves and their inverse fix-ups) run beside them on their own stream instead of between them
                if (!run_has_big_x && !(ctx->sched & 256u)) {
                    // ... and a run WITHOUT a big expansion (a pairing batch of a few checks) keeps them, fix-ups behind them, in its
                    // own chain stream: on one stream for all runs in flight the small expansions - one wave per SIMD each, 0.2-0.35 ms,
                    // plus ~0.1 ms of event latency between them - came to 0.95 ms per run and WERE the step of the 8-GPU shares (device
                    // timeline, exp/dev_timeline.py: a run's expansions started 3.3 ms after its chain had finished).  In the chain
                    // stream a segment's expansion stands in front of the next segment's chain - the run takes ~0.5 ms longer - but a run
                    // in flight costs ONE hardware queue, and what a batch of a few checks needs is runs in flight: beyond ~24 streams in
                    // use the queues are time-sliced (12 runs with an expansion stream each: 1.8 ms per step of 8 bn256 checks; 16 runs
                    // of one stream: 0.64, 2 x bls12_381 0.44).  H2E_SCHED & 512: a stream of the slot (A/B); & 256: the shared small stream
                    if (!(ctx->sched & 512u)) {
                        sx = sa;
                        merged_x = true;
                    } else {
                        if (!J.x_stream) HIP_TRY(make_stream(ctx, &J.x_stream, ctx->prio_expand, 0));
                        sx = J.x_stream;
                        used_x = true;
                    }
                } else {
                    if (!ctx->small_stream) HIP_TRY(make_stream(ctx, &ctx->small_stream, ctx->prio_expand, 0));
                    sx = ctx->small_stream;
                    used_small = true;
                }
            }
            fixup_in_stream = small_x;
            hipEvent_t e = sync_event();
            HIP_TRY(hipEventRecord(e, sa));
            HIP_TRY(hipStreamWaitEvent(sx, e, 0));
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 2), sx)); DBG_STAMP(4 * li + 2, sx);
            // test hook (H2E_OPT_TEST_SKIP_EXPANSION): leave out the expansion of cut segment <si> (or, with -1, of every cut
            // segment but the last): whatever the value chain reads must have been stored by the value chain itself.  The
            // run's status words get H2E_ST_TEST_HOOK, so its arrays cannot be mistaken for a witness.
            bool skip_x = false;
            if (ctx->test_skip_expansion != INT64_MIN) {
                bool later_cut = false;
                for (size_t sj = si + 1; sj < r.segments.size(); sj++) later_cut = later_cut || p->seg_n_sub[sj] > 1;
                skip_x = ctx->test_skip_expansion == (int64_t)si || (ctx->test_skip_expansion == -1 && later_cut);
            }
            if (!skip_x && gate_for < 0) {
                int gpi = next_digit_chain(si);
                if (gpi >= 0) {
                    if (!J.d_gate) {
                        HIP_TRY(hipMalloc((void**)&J.d_gate, 4));
                        HIP_TRY(hipMemset(J.d_gate, 0, 4));
                        J.gate_total = 0;
                    }
                    J.gate_total += n_instances;   // one workgroup per instance
                    gate_for = gpi;
                    int grc = h2e_engine_gate(J.d_gate, J.gate_total, sx);
                    if (grc != 0) return fail(H2E_ERR_HIP, std::string("gate launch failed: ") + hipGetErrorString((hipError_t)grc));
                }
            }
            if (!skip_x && (lrc = expand(sx))) return lrc;
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 3), sx)); DBG_STAMP(4 * li + 3, sx);
            if (!skip_x && (lrc = expand_fixup(sx))) return lrc;
        } else if (p->seg_side_dep[si] != -2) {
            // runs beside the value chain: after the last segment it reads, before the first segment that reads it
            int32_t depi = p->seg_side_dep[si];
            HIP_TRY(hipStreamWaitEvent(se, depi >= 0 && seg_ev[depi] ? seg_ev[depi] : run_begin, 0));
            used_se = true;
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 1), sa)); DBG_STAMP(4 * li + 1, sa);
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 2), se)); DBG_STAMP(4 * li + 2, se);
            if ((lrc = launch(2, se))) return lrc;
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 3), se)); DBG_STAMP(4 * li + 3, se);
            if ((lrc = fixup_after(se))) return lrc;
            side_done[si] = sync_event();
            HIP_TRY(hipEventRecord(side_done[si], se));
        } else if (p->seg_deferrable[si]) {
            // nothing later reads this segment's cells: off the critical stream
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 1), sa)); DBG_STAMP(4 * li + 1, sa);
            hipEvent_t e = sync_event();
            HIP_TRY(hipEventRecord(e, sa));
            HIP_TRY(hipStreamWaitEvent(sb, e, 0));
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 2), sb)); DBG_STAMP(4 * li + 2, sb);
            if ((lrc = launch(2, sb))) return lrc;
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 3), sb)); DBG_STAMP(4 * li + 3, sb);
            if ((lrc = fixup_after(sb))) return lrc;
        } else {
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 1), sa)); DBG_STAMP(4 * li + 1, sa);
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 2), sa)); DBG_STAMP(4 * li + 2, sa);
            if ((lrc = launch(2, sa))) return lrc;
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 3), sa)); DBG_STAMP(4 * li + 3, sa);
            if ((lrc = fixup_after(sa))) return lrc;
        }
        seg_ev[si] = sync_event();
        HIP_TRY(hipEventRecord(seg_ev[si], sa));
        J.n_launches++;
    }
    if (have_pending) {   // (no later segment took it with it)
        int frc = flush_pending();
        if (frc) return frc;
    }
    if (ctx->test_skip_expansion != INT64_MIN) {
        int orc = h2e_engine_or_status(J.d_inst, n_instances, H2E_ST_TEST_HOOK, sa);
        if (orc != 0) return fail(H2E_ERR_HIP, std::string("status kernel launch failed: ") + hipGetErrorString((hipError_t)orc));
    }
    // completion: one stream collects the others and records the slot's `done` event; h2e_run then makes the caller's stream wait for
    // it, h2e_submit leaves that to h2e_wait.  h2e_run: the fix-up stream.  h2e_submit: a stream of the SLOT - on the shared fix-up
    // stream the collectors of all runs in flight stand in submission order, and a run whose slot came free late held back the `done`
    // of every run submitted after it (device timeline of 8-check batches at eight runs in flight, exp/dev_timeline.py: `done` 3-4 ms
    // after the run's last kernel; the slots then came free in bursts and the runs started in bursts)
    {
        hipStream_t sdone = sd;
        if (!join) {
            if (merged_x && !used_x) sdone = sa_main;   // (a run that lives in its chain stream)
            else {
                if (!J.x_stream) HIP_TRY(make_stream(ctx, &J.x_stream, ctx->prio_expand, 0));
                sdone = J.x_stream;
            }
            if (used_sd) {   // the run's fix-ups on the shared fix-up stream
                hipEvent_t e6 = sync_event();
                HIP_TRY(hipEventRecord(e6, sd));
                HIP_TRY(hipStreamWaitEvent(sdone, e6, 0));
            }
        }
        hipEvent_t ea = sync_event();
        HIP_TRY(hipEventRecord(ea, sa_main));
        HIP_TRY(hipStreamWaitEvent(sdone, ea, 0));
        hipEvent_t eb = sync_event();
        HIP_TRY(hipEventRecord(eb, sb));
        HIP_TRY(hipStreamWaitEvent(sdone, eb, 0));
        if (used_se) {
            hipEvent_t e3 = sync_event();
            HIP_TRY(hipEventRecord(e3, se));
            HIP_TRY(hipStreamWaitEvent(sdone, e3, 0));
        }
        if (used_small) {
            hipEvent_t e4 = sync_event();
            HIP_TRY(hipEventRecord(e4, ctx->small_stream));
            HIP_TRY(hipStreamWaitEvent(sdone, e4, 0));
        }
        if (used_x && sdone != J.x_stream) {
            hipEvent_t e5 = sync_event();
            HIP_TRY(hipEventRecord(e5, J.x_stream));
            HIP_TRY(hipStreamWaitEvent(sdone, e5, 0));
        }
        if (d_digests) {   // every kernel that adds to the digest shards has finished here
            int drc = h2e_engine_digest_reduce(J.dg_shards, H2E_DG_SHARDS, 3 * n_instances * 4, d_digests, sdone);
            if (drc != 0) return fail(H2E_ERR_HIP, std::string("digest kernel launch failed: ") + hipGetErrorString((hipError_t)drc));
        }
        HIP_TRY(hipEventRecord(J.done, sdone));
        DBG_STAMP(1000u, sdone);
        if (join) HIP_TRY(hipStreamWaitEvent(sa_main, J.done, 0));
    }
#ifdef H2E_DEBUG_HOOKS
    dbg_mark("queued");
#endif
    return 0;
}

int h2e_run(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
            void* d_select, void* d_status, void* stream_) {
    if (n_instances == 0) return 0;
    return run_impl(ctx, p, n_instances, d_inputs, d_base, d_range, d_select, d_status, (hipStream_t)stream_, true, nullptr);
}

// h2e_run with halo2's advice columns coming straight out of the expansion (no h2e_export pass): see include/h2e.h
int h2e_run_columns(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range, void* d_select,
                    void* d_cols_base, void* d_cols_range, void* d_cols_select, int form, void* d_status, void* stream_) {
    if (n_instances == 0) return 0;
    if (!d_cols_base || !d_cols_range || !d_cols_select) return fail(H2E_ERR_INVALID, "null column array");
    if (form != H2E_FORM_CANONICAL) return fail(H2E_ERR_INVALID, "h2e_run_columns: canonical cells only so far (H2E_FORM_MONTGOMERY: h2e_export)");
    RunColumns rc;
    rc.col[0] = d_cols_base; rc.col[1] = d_cols_range; rc.col[2] = d_cols_select;
    rc.form = form;
    return run_impl(ctx, p, n_instances, d_inputs, d_base, d_range, d_select, d_status, (hipStream_t)stream_, true, nullptr, nullptr, nullptr, nullptr, &rc);
}

int h2e_submit(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
               void* d_select, void* d_status, void* stream_, int* job) {
    if (!job) return fail(H2E_ERR_INVALID, "job is null");
    *job = -1;
    if (n_instances == 0) return fail(H2E_ERR_INVALID, "n_instances must be > 0");
    return run_impl(ctx, p, n_instances, d_inputs, d_base, d_range, d_select, d_status, (hipStream_t)stream_, false, job);
}

int h2e_run_digest(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
                   void* d_select, void* d_status, void* d_digests, void* stream_) {
    if (n_instances == 0) return 0;
    if (!d_digests) return fail(H2E_ERR_INVALID, "d_digests is null");
    return run_impl(ctx, p, n_instances, d_inputs, d_base, d_range, d_select, d_status, (hipStream_t)stream_, true, nullptr, d_digests);
}
int h2e_submit_digest(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
                      void* d_select, void* d_status, void* d_digests, void* stream_, int* job) {
    if (!job) return fail(H2E_ERR_INVALID, "job is null");
    *job = -1;
    if (n_instances == 0) return fail(H2E_ERR_INVALID, "n_instances must be > 0");
    if (!d_digests) return fail(H2E_ERR_INVALID, "d_digests is null");
    return run_impl(ctx, p, n_instances, d_inputs, d_base, d_range, d_select, d_status, (hipStream_t)stream_, false, job, d_digests);
}

// Several caller batches as ONE run.  What a stream of small batches costs is runs, not instances: a pairing check's value chain is ~2 ms
// of latency on one compute unit whatever the batch (a 2-check bls12_381 batch 0.20 ms per check pipelined, a 16-check batch 0.07), and
// streams in use beyond ~24 are time-sliced, so more runs in flight are not to be had.  Eight 2-check batches submitted together execute
// as one 16-instance run - same kernels, same launches - and every batch's cells land in its own arrays.
static int check_batches(uint32_t n_batches, uint32_t n_each, const void* const* d_inputs, void* const* d_base, void* const* d_range,
                         void* const* d_select, void* const* d_status) {
    if (n_batches == 0 || n_batches > 16) return fail(H2E_ERR_INVALID, "n_batches must be 1 .. 16");
    if (n_each == 0) return fail(H2E_ERR_INVALID, "n_instances_each must be > 0");
    if (!d_inputs || !d_base || !d_range || !d_select || !d_status) return fail(H2E_ERR_INVALID, "null pointer table");
    for (uint32_t b = 0; b < n_batches; b++) {
        if (!d_inputs[b] || !d_base[b] || !d_range[b] || !d_select[b] || !d_status[b]) return fail(H2E_ERR_INVALID, "null device pointer in a batch");
        for (uint32_t c = 0; c < b; c++)
            if (d_base[b] == d_base[c] || d_range[b] == d_range[c] || d_select[b] == d_select[c] || d_status[b] == d_status[c])
                return fail(H2E_ERR_INVALID, "two batches of a run share an output array");
    }
    return 0;
}
int h2e_run_batches(h2e_ctx* ctx, h2e_program* p, uint32_t n_batches, uint32_t n_instances_each, const void* const* d_inputs,
                    void* const* d_base, void* const* d_range, void* const* d_select, void* const* d_status, void* stream_) {
    int rc = check_batches(n_batches, n_instances_each, d_inputs, d_base, d_range, d_select, d_status);
    if (rc) return rc;
    RunBatches bs;
    bs.k = n_batches; bs.arr_n = n_instances_each; bs.inputs = d_inputs; bs.base = d_base; bs.range = d_range; bs.select = d_select; bs.status = d_status;
    return run_impl(ctx, p, n_batches * n_instances_each, nullptr, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream_, true, nullptr, nullptr, &bs);
}
int h2e_submit_batches(h2e_ctx* ctx, h2e_program* p, uint32_t n_batches, uint32_t n_instances_each, const void* const* d_inputs,
                       void* const* d_base, void* const* d_range, void* const* d_select, void* const* d_status, void* stream_, int* job) {
    if (!job) return fail(H2E_ERR_INVALID, "job is null");
    *job = -1;
    int rc = check_batches(n_batches, n_instances_each, d_inputs, d_base, d_range, d_select, d_status);
    if (rc) return rc;
    RunBatches bs;
    bs.k = n_batches; bs.arr_n = n_instances_each; bs.inputs = d_inputs; bs.base = d_base; bs.range = d_range; bs.select = d_select; bs.status = d_status;
    return run_impl(ctx, p, n_batches * n_instances_each, nullptr, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream_, false, job, nullptr, &bs);
}

int h2e_wait(h2e_ctx* ctx, int job, void* stream_) {
    if (!ctx) return fail(H2E_ERR_INVALID, "null ctx");
    std::lock_guard<std::mutex> guard(ctx->mu);   // (the slot's event is created under this lock by run_impl)
    if (job < 0 || job >= (int)ctx->depth || !ctx->slots[job].done) return fail(H2E_ERR_INVALID, "bad job");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream_, ctx->slots[job].done, 0));
    return 0;
}

