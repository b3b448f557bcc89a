// h2e_program: where segments run (deferrable / side segments), the order tables' sub-range slots, and finish(): the passes in order.
// Part of the C-ABI layer's one translation unit (included by h2e_capi.cpp).
#pragma once

// A segment without cuts normally runs on the caller's (critical) stream because later value-chain kernels may
// read any of its cells.  If no reference anywhere (later ops, strand parameters, candidate tables, predictor
// arguments, outputs) points into its rows, it can run on the expansion stream instead.
void h2e_program::mark_deferrable() {
    h2e::Recorder& r = *rec;
    seg_deferrable.assign(r.segments.size(), 0);
    for (size_t si = 0; si < r.segments.size(); si++) {
        const h2e::Segment& sg = r.segments[si];
        if (sg.n_cuts != 0 || sg.tape_end <= sg.tape_begin || !sg.is_fork) continue;
        uint32_t lo[3] = {sg.base0, sg.range0, sg.select0};
        uint64_t hi[3] = {sg.base0 + (uint64_t)sg.dbase * sg.n_strands, sg.range0 + (uint64_t)sg.drange * sg.n_strands,
                          sg.select0 + (uint64_t)sg.dselect * sg.n_strands};
        auto hits = [&](uint32_t ref) {
            if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM || H2E_REF_REL(ref)) return false;
            uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
            return row >= lo[region] && row < hi[region];
        };
        bool referenced = false;
        for (size_t sj = 0; sj < r.segments.size() && !referenced; sj++) {
            if (sj == si) continue;
            for (uint32_t i = r.segments[sj].tape_begin; i < r.segments[sj].tape_end && !referenced; i++)
                for (int k = 0; k < H2E_OP_MAX_REFS; k++) referenced = referenced || hits(r.tape[i].refs[k]);
        }
        bool dbg = dbg_env("H2E_DUMP_TAPE") != nullptr;
        if (dbg && referenced) fprintf(stderr, "segment %zu referenced by ops\n", si);
        for (uint32_t ref : r.aux) if (hits(ref)) { if (dbg && !referenced) fprintf(stderr, "segment %zu referenced by aux %08x\n", si, ref); referenced = true; }
        for (uint32_t ref : r.params) if (hits(ref)) { if (dbg && !referenced) fprintf(stderr, "segment %zu referenced by params %08x\n", si, ref); referenced = true; }
        for (uint32_t ref : r.outputs) referenced = referenced || hits(ref);
        for (uint32_t ref : r.pre_args) if (hits(ref)) { if (dbg && !referenced) fprintf(stderr, "segment %zu referenced by pre_args %08x\n", si, ref); referenced = true; }
        seg_deferrable[si] = referenced ? 0 : 1;
        if (dbg_env("H2E_DUMP_TAPE")) fprintf(stderr, "segment %zu deferrable %d\n", si, (int)seg_deferrable[si]);
    }
}

void h2e_program::mark_side_segments() {
    h2e::Recorder& r = *rec;
    size_t ns = r.segments.size();
    seg_side_dep.assign(ns, -2);
    seg_first_reader.assign(ns, (uint32_t)ns);
    // first rows of every non-empty segment, per region (rows are handed out in program order)
    std::vector<uint32_t> ids;
    std::vector<std::array<uint32_t, 3>> start;
    for (size_t si = 0; si < ns; si++) {
        const h2e::Segment& sg = r.segments[si];
        if (sg.tape_end <= sg.tape_begin) continue;
        std::array<uint32_t, 3> st;
        if (sg.is_fork) st = {sg.base0, sg.range0, sg.select0};
        else st = {r.tape[sg.tape_begin].base_row, r.tape[sg.tape_begin].range_row, r.tape[sg.tape_begin].select_row};
        ids.push_back((uint32_t)si);
        start.push_back(st);
    }
    auto segment_of = [&](uint32_t ref) -> int {
        if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM || H2E_REF_REL(ref)) return -1;
        uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
        int ans = -1;
        for (size_t k = 0; k < ids.size(); k++)
            if (start[k][region] <= row) ans = (int)ids[k];
        return ans;
    };
    for (size_t si = 0; si < ns; si++) {
        const h2e::Segment& sg = r.segments[si];
        if (sg.n_cuts != 0 || sg.tape_end <= sg.tape_begin || !sg.is_fork) continue;
        bool has_pre = false;
        for (auto& pk : r.pre_kernels) has_pre = has_pre || pk.before_segment == si;
        if (has_pre) continue;
        int last_dep = -1;
        bool ok = true;
        auto dep = [&](uint32_t ref) {
            int sj = segment_of(ref);
            if (sj < 0) return;
            if ((size_t)sj >= si) { ok = false; return; }
            if (r.segments[sj].n_cuts != 0) ok = false;
            if (seg_side_dep[sj] != -2) return;   // another side segment: the side stream runs them in order
            last_dep = std::max(last_dep, sj);
        };
        for (uint32_t i = sg.tape_begin; i < sg.tape_end; i++)
            for (int k = 0; k < H2E_OP_MAX_REFS; k++) dep(r.tape[i].refs[k]);
        for (size_t q = 0; q < (size_t)sg.n_params * sg.n_strands; q++)
            if (sg.params_begin + q < r.params.size()) dep(r.params[sg.params_begin + q]);
        if (!ok) continue;
        // first later reader
        uint32_t lo[3] = {sg.base0, sg.range0, sg.select0};
        uint64_t hi[3] = {sg.base0 + (uint64_t)sg.dbase * sg.n_strands, sg.range0 + (uint64_t)sg.drange * sg.n_strands,
                          sg.select0 + (uint64_t)sg.dselect * sg.n_strands};
        auto hits = [&](uint32_t ref) {
            if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM || H2E_REF_REL(ref)) return false;
            uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
            return row >= lo[region] && row < hi[region];
        };
        uint32_t first_reader = (uint32_t)ns;
        for (size_t sj = si + 1; sj < ns && first_reader == ns; sj++) {
            const h2e::Segment& sr = r.segments[sj];
            bool reads = false;
            for (uint32_t i = sr.tape_begin; i < sr.tape_end && !reads; i++)
                for (int k = 0; k < H2E_OP_MAX_REFS; k++) reads = reads || hits(r.tape[i].refs[k]);
            for (size_t q = 0; q < (size_t)sr.n_params * sr.n_strands && !reads; q++)
                if (sr.params_begin + q < r.params.size()) reads = hits(r.params[sr.params_begin + q]);
            for (auto& pk : r.pre_kernels) {
                if (pk.before_segment != sj || reads) continue;
                for (uint32_t q = 0; q < pk.k.n_params * pk.k.n_lanes && !reads; q++)
                    if (pk.k.params_begin + q < r.params.size()) reads = hits(r.params[pk.k.params_begin + q]);
                reads = reads || true;   // predictor arguments are not delimited per kernel: be conservative
            }
            if (reads) first_reader = (uint32_t)sj;
        }
        bool in_aux = false;
        for (uint32_t ref : r.aux) in_aux = in_aux || hits(ref);
        if (in_aux) first_reader = std::min<uint32_t>(first_reader, (uint32_t)si + 1);
        if (first_reader <= si + 1) continue;   // nothing to overlap with
        seg_side_dep[si] = last_dep;
        seg_first_reader[si] = first_reader;
        if (dbg_env("H2E_DUMP_TAPE")) fprintf(stderr, "segment %zu: side stream after segment %d, first reader %u\n", si, last_dep, first_reader);
    }
}

// Expansion result cache (engine.hip ld_int_x / xc_put_x): per sub-range of a cut segment, which of the three LDS
// entries an integer result goes to and which operands are read from them - furthest-next-use replacement over the
// static op sequence.  Encoded in op.flags bits 8-15.
void h2e_program::assign_expansion_slots() {
    h2e::Recorder& r = *rec;
    const int L = r.fp.limbs;
    const int NSLOT = 3;
    for (auto& sg : r.segments) {
        uint32_t n_ops = sg.tape_end - sg.tape_begin;
        if (sg.n_cuts == 0 || n_ops == 0) continue;
        H2EOp* ops = r.tape.data() + sg.tape_begin;
        const uint32_t rel = sg.is_fork ? 1 : 0;
        std::vector<uint32_t> bounds;
        uint32_t lastb = 0;
        for (uint32_t k = 0; k < sg.n_cuts; k++) {
            uint32_t at = r.cuts[sg.cuts_begin + k];
            if (at > lastb && at < n_ops) {
                bounds.push_back(at);
                lastb = at;
            }
        }
        bounds.push_back(n_ops);
        auto result_key = [&](const H2EOp& op) -> uint32_t {
            switch (op.opcode) {
                case H2E_OP_INT_ADD: case H2E_OP_INT_SUB: case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL:
                    return H2E_MAKE_REF(0, 4, rel, op.base_row);
                case H2E_OP_INT_MUL: case H2E_OP_REDUCE: case H2E_OP_DIV_CORE:
                    return H2E_MAKE_REF(1, 0, rel, op.range_row);
                default: return H2E_NO_REF;
            }
        };
        auto operand_pos = [&](const H2EOp& op, int* pos) -> int {
            switch (op.opcode) {
                case H2E_OP_INT_ADD: case H2E_OP_INT_SUB: case H2E_OP_INT_MUL: case H2E_OP_DIV_CORE:
                    pos[0] = 0; pos[1] = L + 1; return 2;
                case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: case H2E_OP_REDUCE: case H2E_OP_IS_INT_ZERO: case H2E_OP_MASK_INT:
                    pos[0] = 0; return 1;
                default: return 0;
            }
        };
        uint32_t lo = 0;
        for (uint32_t hi : bounds) {
            std::map<uint32_t, uint32_t> producer_of;          // key -> op index
            std::map<uint32_t, std::vector<uint32_t>> uses;     // producer op -> consumer op indices (ascending)
            for (uint32_t i = lo; i < hi; i++) {
                ops[i].flags &= 0x00ff;
                int pos[2];
                int n = operand_pos(ops[i], pos);
                for (int q = 0; q < n; q++) {
                    auto it = producer_of.find(ops[i].refs[pos[q]]);
                    if (it != producer_of.end()) uses[it->second].push_back(i);
                }
                uint32_t key = result_key(ops[i]);
                if (key != H2E_NO_REF) producer_of[key] = i;
            }
            int owner[NSLOT];
            for (int sl = 0; sl < NSLOT; sl++) owner[sl] = -1;
            std::map<uint32_t, size_t> next;   // producer -> index of its next unconsumed use
            auto next_use = [&](uint32_t p) -> uint32_t {
                auto& u = uses[p];
                size_t k = next[p];
                return k < u.size() ? u[k] : 0xffffffffu;
            };
            std::map<uint32_t, int> slot_of;
            for (uint32_t i = lo; i < hi; i++) {
                int pos[2];
                int n = operand_pos(ops[i], pos);
                for (int q = 0; q < n; q++) {
                    auto it = producer_of.find(ops[i].refs[pos[q]]);
                    if (it == producer_of.end() || it->second >= i) continue;
                    uint32_t pr = it->second;
                    // (a key can be produced twice in a sub-range only if rows repeated, which they do not)
                    auto st = slot_of.find(pr);
                    if (st != slot_of.end() && owner[st->second] == (int)pr) ops[i].flags |= (uint16_t)((st->second + 1) << (10 + 2 * q));
                    auto& u = uses[pr];
                    while (next[pr] < u.size() && u[next[pr]] <= i) next[pr]++;
                }
                for (int sl = 0; sl < NSLOT; sl++)
                    if (owner[sl] >= 0 && next_use((uint32_t)owner[sl]) == 0xffffffffu) owner[sl] = -1;
                if (result_key(ops[i]) != H2E_NO_REF && !uses[i].empty()) {
                    int pick = -1;
                    for (int sl = 0; sl < NSLOT && pick < 0; sl++)
                        if (owner[sl] < 0) pick = sl;
                    if (pick < 0) {
                        int far = 0;
                        for (int sl = 1; sl < NSLOT; sl++)
                            if (next_use((uint32_t)owner[sl]) > next_use((uint32_t)owner[far])) far = sl;
                        if (next_use((uint32_t)owner[far]) > uses[i][0]) pick = far;
                    }
                    if (pick >= 0) {
                        owner[pick] = (int)i;
                        slot_of[i] = pick;
                        ops[i].flags |= (uint16_t)((pick + 1) << 8);
                    }
                }
            }
            lo = hi;
        }
    }
}

void h2e_program::finish() {
    h2e::Recorder& r = *rec;
    if (r.fp.id != r.primary_field) r.use_field(r.primary_field);   // the analyses below decode cut segments in the program's field
    r.close_segment();
    mark_local_results();
    assign_expansion_slots();
    mark_deferrable();
    mark_side_segments();
    // The serial tail of the program: the last cut single-strand segment whose predictor chain already starts early
    // on the side stream (it only needs an earlier segment's predictors - the MSM tail), provided nothing after it
    // forks again.  Its whole value chain, and whatever follows it, runs on the job slot's side stream (run_impl).
    tail_from = -1;
    for (size_t si = 0; si < r.segments.size(); si++) {
        const h2e::Segment& sg = r.segments[si];
        if (sg.tape_end <= sg.tape_begin || sg.n_strands != 1 || sg.n_cuts == 0) continue;
        bool early_chain = false;
        for (auto& pk : r.pre_kernels) early_chain = early_chain || (pk.before_segment == si && pk.early_after_segment >= 0);
        bool forks_later = false;
        for (size_t sj = si + 1; sj < r.segments.size(); sj++)
            forks_later = forks_later || (r.segments[sj].tape_end > r.segments[sj].tape_begin && r.segments[sj].n_strands > 1);
        if (early_chain && !forks_later) {
            tail_from = (int64_t)si;
            break;
        }
    }
    // which of the 8 value-hint slots per ecc op does anything read?  (finalize_ecc skips the others)
    for (auto& pk : r.pre_kernels) {
        if (!pk.k.ecc_ops) continue;
        uint32_t lo = pk.k.hint_base, per = pk.k.hints_per_lane, mask = 1u << H2E_HINT_LAMBDA;
        for (const H2EOp& op : r.tape)
            if ((op.flags & H2E_FLAG_HINTED) && op.imm >= lo && op.imm < lo + per) mask |= 1u << ((op.imm - lo) % H2E_ECC_HINT_SLOTS);
        pk.k.used_slots = mask;
        if (dbg_env("H2E_DUMP_TAPE")) fprintf(stderr, "predictor kind %u: value-hint slots in use: 0x%02x\n", pk.k.kind, mask);
    }
    if (dbg_env("H2E_DUMP_TAPE")) {   // debugging aid: per segment, ops by opcode (alive / skipped by the values replay)
        for (size_t si = 0; si < r.segments.size(); si++) {
            auto& sg = r.segments[si];
            std::map<int, std::array<uint32_t, 4>> h;
            for (uint32_t i = sg.tape_begin; i < sg.tape_end; i++) {
                auto& e = h[r.tape[i].opcode];
                e[(r.tape[i].flags & H2E_FLAG_VALUES_SKIP) ? 1 : 0]++;
                if (!(r.tape[i].flags & H2E_FLAG_VALUES_SKIP) && (r.tape[i].flags & H2E_FLAG_HINTED)) e[2]++;
                if (!(r.tape[i].flags & H2E_FLAG_VALUES_SKIP) && !(r.tape[i].flags & H2E_FLAG_LOCAL_RESULT)) e[3]++;
            }
            fprintf(stderr, "segment %zu: ops %u strands %u cuts %u fork %d\n", si, sg.tape_end - sg.tape_begin, sg.n_strands, sg.n_cuts, (int)sg.is_fork);
            for (auto& kv : h)
                fprintf(stderr, "   opcode %2d alive %6u (hinted %6u, stored %6u) skipped %6u\n", kv.first, kv.second[0], kv.second[2], kv.second[3], kv.second[1]);
        }
    }
    base_rows = std::max<uint64_t>(r.base_height, r.base_offset) + 1;
    range_rows = std::max<uint64_t>(r.range_height, r.range_offset) + 1;
    select_rows = std::max<uint64_t>(r.select_height, r.select_offset) + 1;
    if (r.emit_shape) {
        r.base_fix.resize(base_rows * 9, 0);
        r.range_fix.resize(range_rows * 2, 0);
        r.select_fix.resize(select_rows * 2, 0);
        r.base_flags.resize(base_rows * 5, 0);
        r.range_flags.resize(range_rows * 3, 0);
        r.select_flags.resize(select_rows * 2, 0);
        perm_flat.reserve(r.permutations.size() * 2);
        for (auto& p : r.permutations) {
            perm_flat.push_back(p.first);
            perm_flat.push_back(p.second);
        }
        for (auto& f : r.fixed_patches) {
            patch_flat.push_back(f.row);
            patch_flat.push_back(f.col);
            patch_flat.push_back(f.input_slot);
            patch_flat.push_back((uint32_t)f.limb);
        }
    }
}

