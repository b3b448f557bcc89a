// What a context and a run in flight own: job slots (workspace, instance table, events, streams) and struct h2e_ctx.
// Part of the C-ABI layer's one translation unit (included by h2e_capi.cpp).
#pragma once

// Everything one run owns while it is in flight: engine workspace, instance table, events.  A context keeps a small
// ring of these, so that h2e_submit can queue the value chain of run k + 1 (caller's stream) while run k's expansion is
// still streaming on the expansion stream; h2e_run uses the same slots and joins before it returns.
#define H2E_DG_SHARDS 64u
struct JobSlot {
    // engine workspace (grow-only): quotient hints, numerator/denominator pairs, Jacobian scratch, selected points
    uint64_t *ws_hints = nullptr, *ws_nd = nullptr, *ws_jac = nullptr, *ws_sel = nullptr;
    size_t ws_hints_words = 0, ws_nd_words = 0, ws_jac_words = 0, ws_sel_words = 0;
    InstanceDescHost* d_inst = nullptr;   // the run's table of per-instance descriptors (written on the device: handoff.hip)
    uint32_t inst_cap = 0;
    uint64_t* dg_shards = nullptr;    // stream digest accumulators of the slot's run: [H2E_DG_SHARDS][3][instances][4] words
    uint32_t dg_cap = 0;              // instances they are sized for
    std::vector<hipEvent_t> ev;       // profiling: 4 per launched segment (value-chain begin/end, expansion begin/end)
    std::vector<hipEvent_t> sync_ev;  // cross-stream dependencies
    hipEvent_t done = nullptr;        // recorded when every stream of the slot's last run has finished
    hipEvent_t order_ev = nullptr;    // the caller's stream at submission (h2e_submit)
    // Side stream of the slot: early predictors, fork segments outside the chain, and the *serial tail* of a program
    // (the MSM tail: a single-wave 13 ms predictor chain + its replay).  Per slot, so that the tail of run k and the
    // value chain of run k + 1 (caller's stream) run side by side - the tail chain is latency-, not throughput-bound.
    hipStream_t side_stream = nullptr;
    // Chain stream of the slot (h2e_submit only): the run's value chain is queued here, ordered after what the caller's
    // stream held at submission, so that the value chains of consecutive runs overlap each other as well - each of their
    // kernels is latency-bound and leaves most of the GPU idle.
    hipStream_t chain_stream = nullptr;
    // completion stream of the slot (h2e_submit): collects the run's streams and records `done` (run.hpp) - unless the run lives in its
    // chain stream alone (a pipelined run without a big expansion)
    hipStream_t x_stream = nullptr;
    // start counter of the slot's digit chains and what it will read once every chain launched so far has started (engine.hip h2e_gate)
    uint32_t* d_gate = nullptr;
    uint32_t gate_total = 0;
    bool used = false;
    bool profiled = false;            // the last run on this slot recorded `ev`
    uint32_t n_launches = 0;
    std::vector<uint32_t> x_kernels;  // per launched segment: expansion kernel launches of the last run (2 = split)
    void release() {
        for (auto e : ev) (void)hipEventDestroy(e);
        for (auto e : sync_ev) (void)hipEventDestroy(e);
        if (done) (void)hipEventDestroy(done);
        if (order_ev) (void)hipEventDestroy(order_ev);
        if (side_stream) (void)hipStreamDestroy(side_stream);
        if (chain_stream) (void)hipStreamDestroy(chain_stream);
        if (x_stream) (void)hipStreamDestroy(x_stream);
        (void)hipFree(ws_hints);
        (void)hipFree(ws_nd);
        (void)hipFree(ws_jac);
        (void)hipFree(ws_sel);
        (void)hipFree(d_inst);
        (void)hipFree(dg_shards);
        (void)hipFree(d_gate);
    }
};

struct h2e_ctx {
    int device;
    H2EFieldConsts* d_fc[3] = {nullptr, nullptr, nullptr};
    std::map<std::string, h2e_program*> cache;
    bool profiling = false;
    bool cols_consts = false;   // the column-emission unit has its field constants (its own constant memory)
    static constexpr int N_SLOTS = 32;
    uint32_t depth = 2;      // job slots in use = runs in flight (H2E_OPT_PIPELINE_DEPTH); each slot brings its own streams: 2 hide an MSM
                             // step's value chain, a pairing batch of a few checks wants 16 (its chains are latency-bound on one CU per check)
    JobSlot slots[N_SLOTS];
    uint64_t n_runs = 0;     // runs submitted so far: run k uses slot k % N_SLOTS
    int last_slot = -1;
    hipStream_t expand_stream = nullptr;
    hipStream_t expand_stream2 = nullptr;   // the big (not huge) expansions of odd job slots (run.hpp)
    hipStream_t fixup_stream = nullptr;
    hipStream_t small_stream = nullptr;   // small expansions of pipelined runs (H2E_SCHED & 4)
    // tuning knobs, read once at h2e_ctx_create (H2E_X_SPLIT, H2E_X_SPLIT_MIN_LANES); h2e_ctx_set_option overrides
    uint32_t x_split_pct = 45;
    // launches a big expansion goes out as (H2E_X_PARTS): the part behind the first x_split_pct percent in parts - 1 equal launches.  The last
    // part's inverse fix-up is the one nothing runs under, and the run is complete - its buffer set free for the run after the next - only
    // behind it: 64 x 1024-point tiles pipelined, alternating in one box: 2 launches 15.38 / 15.43 ms per step, 3: 15.13 / 15.15, 4: 15.11 / 15.18
    uint32_t x_parts = 3;
    uint64_t x_split_min_lanes = 1ull << 21;
    uint64_t small_x_lanes = 1u << 18;   // an expansion with fewer lanes is "small" (H2E_SMALL_X_LANES)
    uint32_t sched = 4;      // (bit 8, round 6: a pipelined run's held-back expansion - the MSM's candidate tables - beside the other runs' big expansions on
                             // the small-expansion stream instead of between them: 14.65 -> 14.45 ms per step with three runs in flight, alternating in
                             // one box - but the window launches then share the store stream and their own rate, the record's roofline figure,
                             // falls from 0.66 to 0.59 of the roof: left off)
                             // scheduling experiments (H2E_SCHED bit mask): 1 = a pipelined run's small fix-ups go to the slot's side
                             // stream, 2 = its small expansions too (instead of queueing on the shared expansion stream)
    int prio_expand = 0, prio_side = 0, prio_fixup = 0;   // HIP stream priorities (H2E_STREAM_PRIORITIES="x,s,f"; lower = higher priority)
    int64_t test_skip_expansion = INT64_MIN;   // test hook (h2e_ctx_set_option): see H2E_OPT_TEST_SKIP_EXPANSION
    uint32_t last_split_segments = 0;          // segments of the last run whose expansion was split (h2e_ctx_get_stat)
    std::mutex mu;                             // h2e_run / h2e_submit on one context are serialised on the host
    // Operator API: programs of the ops recorded so far, keyed by (op, arguments, operand handles, cursors, heights, msm prefix):
    // a records object that repeats an op sequence (the next batch of the same circuit) re-uses them - no host-side recording,
    // no new device tapes.  `outs` = the handles the op returned, byte for byte.
    // The cache is bounded (keys hold value-dependent arguments - constants, offsets - so a long-lived context would otherwise
    // keep one program with its device tapes per distinct key): at `op_cache_cap` entries the least recently used ones that no
    // call is running go (H2E_OP_CACHE_CAP, default 4096; a proving loop's working set is its ops per batch).
    struct OpEntry {
        h2e_program* prog = nullptr;
        std::vector<std::vector<uint8_t>> outs;
        size_t msm_prefix_after = 0;
        uint64_t last_use = 0;
        uint32_t in_use = 0;
    };
    std::map<std::string, OpEntry> op_cache;
    std::mutex op_mu;
    uint64_t op_hits = 0, op_misses = 0, op_tick = 0, op_evictions = 0;
    size_t op_cache_cap = 4096;
    void op_cache_trim() {   // (op_mu held)
        while (op_cache.size() > op_cache_cap) {
            auto victim = op_cache.end();
            for (auto it = op_cache.begin(); it != op_cache.end(); ++it)
                if (it->second.in_use == 0 && (victim == op_cache.end() || it->second.last_use < victim->second.last_use)) victim = it;
            if (victim == op_cache.end()) break;
            delete victim->second.prog;   // (frees its device tapes: hipFree waits for the work that still reads them)
            op_cache.erase(victim);
            op_evictions++;
        }
    }
    ~h2e_ctx() {
        for (auto& kv : op_cache) delete kv.second.prog;
        for (auto& kv : cache) delete kv.second;
        for (int i = 0; i < 3; i++)
            if (d_fc[i]) (void)hipFree(d_fc[i]);
        for (auto& sl : slots) sl.release();
        if (expand_stream) (void)hipStreamDestroy(expand_stream);
        if (expand_stream2) (void)hipStreamDestroy(expand_stream2);
        if (fixup_stream) (void)hipStreamDestroy(fixup_stream);
        if (small_stream) (void)hipStreamDestroy(small_stream);
    }
};

