// h2e_program: the values-only replay compiler of a cut segment - V-tape records, LDS slots by liveness, restartable pieces, staged
// inputs, the level schedule of chains without hints.  Part of the C-ABI layer's one translation unit (included by h2e_capi.cpp).
#pragma once

// Compile one cut segment into V-tape records (tape.h).  Values = results of alive ops; each gets an LDS slot for
// as long as later ops of the replay read it (furthest-next-use eviction when the slots run out: an evicted or
// never cached value goes through its cells, so its producer stores it).
void h2e_program::compile_replay(const h2e::Segment* sg, const H2EOp* ops, uint32_t n_ops, const uint32_t* first, const uint32_t* last,
                    const std::function<int(uint32_t, uint32_t)>& producer) {
    h2e::Recorder& r = *rec;
    const int L = r.fp.limbs;
    const uint32_t NS = L == 3 ? 22 : 18, NF = 4;   // VSlots in engine.hip (NF) and the LDS budget (NS)
    const uint32_t rel = sg->is_fork ? 1 : 0;
    enum { K_NONE, K_MUL, K_ADD, K_FE, K_SEL, K_FULL, K_CONST };
    auto kind_of = [](const H2EOp& op) -> int {
        switch (op.opcode) {
            case H2E_OP_INT_MUL: case H2E_OP_REDUCE: case H2E_OP_DIV_CORE: return K_MUL;
            case H2E_OP_INT_ADD: case H2E_OP_INT_SUB: case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: case H2E_OP_MASK_INT:
            case H2E_OP_BISEC_INT: return K_ADD;
            case H2E_OP_IS_INT_ZERO: case H2E_OP_NOT: case H2E_OP_AND: case H2E_OP_OR: case H2E_OP_XNOR: case H2E_OP_PICK_INDEX:
                return K_FE;
            case H2E_OP_SELECT_POINT: return K_SEL;
            case H2E_OP_CONST_INT: return K_CONST;   // a constant of the pool: a value like any other (cells: column 0)
            case H2E_OP_ASSERT_CONST: case H2E_OP_CACHE_INT: case H2E_OP_SUM_LIMBS: case H2E_OP_NOP: return K_NONE;
            default: return K_FULL;
        }
    };
    auto fe_row = [&](const H2EOp& op) -> uint32_t {
        if (op.opcode == H2E_OP_IS_INT_ZERO) return op.base_row + 6 + 4 * (uint32_t)r.fp.pure_w_check_limbs;
        if (op.opcode == H2E_OP_PICK_INDEX) return op.base_row + (op.imm < 5 ? 0 : 1);
        return op.base_row;
    };
    struct Opd { uint32_t ref; bool is_int; int refpos; };
    auto operands = [&](const H2EOp& op, Opd* o) -> int {
        bool hinted = (op.flags & H2E_FLAG_HINTED) != 0;
        int n = 0;
        switch (op.opcode) {
            case H2E_OP_INT_MUL: case H2E_OP_DIV_CORE:
                if (!hinted) { o[n++] = {op.refs[0], true, 0}; o[n++] = {op.refs[L + 1], true, L + 1}; }
                break;
            case H2E_OP_REDUCE:
                if (!hinted) o[n++] = {op.refs[0], true, 0};
                break;
            case H2E_OP_INT_ADD: case H2E_OP_INT_SUB:
                o[n++] = {op.refs[0], true, 0}; o[n++] = {op.refs[L + 1], true, L + 1};
                break;
            case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: case H2E_OP_IS_INT_ZERO:
                o[n++] = {op.refs[0], true, 0};
                break;
            case H2E_OP_MASK_INT:
                o[n++] = {op.refs[0], true, 0}; o[n++] = {op.refs[L + 1], false, L + 1};
                break;
            case H2E_OP_BISEC_INT:
                o[n++] = {op.refs[0], false, 0}; o[n++] = {op.refs[1], true, 1}; o[n++] = {op.refs[L + 2], true, L + 2};
                break;
            case H2E_OP_SELECT_POINT:
                if (!(op.flags & H2E_FLAG_PRESELECTED)) o[n++] = {op.refs[0], false, 0};
                break;
            case H2E_OP_NOT:
                o[n++] = {op.refs[0], false, 0};
                break;
            case H2E_OP_AND: case H2E_OP_OR: case H2E_OP_XNOR:
                o[n++] = {op.refs[0], false, 0}; o[n++] = {op.refs[1], false, 1};
                break;
            default: break;
        }
        return n;
    };
    // value id = 2 * op + which; -1: not a value of this replay (read from its cell)
    auto value_of = [&](uint32_t ref, bool is_int) -> int {
        if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM || H2E_REF_REL(ref) != rel) return -1;
        uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref), col = H2E_REF_COL(ref);
        if (!rel && (row < first[region] || row >= last[region])) return -1;
        int p = producer(region, row);
        if (p < 0) return -1;
        const H2EOp& po = ops[p];
        int pk = kind_of(po);
        if (pk == K_FULL) return -1;   // its rows are written for real
        if (is_int) {
            if (pk == K_MUL && region == 1 && col == 0 && row == po.range_row) return 2 * p;
            if (pk == K_ADD && region == 0 && col == 4 && row == po.base_row) return 2 * p;
            if (pk == K_CONST && region == 0 && col == 0 && row == po.base_row) return 2 * p;
            if (pk == K_SEL && region == 2 && col == 0 && row == po.select_row) return 2 * p;
            if (pk == K_SEL && region == 2 && col == 0 && row == po.select_row + (uint32_t)L + 1) return 2 * p + 1;
        } else {
            if (pk == K_FE && region == 0 && col == 4 && row == fe_row(po)) return 2 * p;
        }
        return -2;   // a cell the values-only replay never writes
    };
    // the op of this segment (index into ops) that writes a referenced cell, or -1
    auto writer_of = [&](uint32_t ref) -> int {
        if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM || H2E_REF_REL(ref) != rel) return -1;
        uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
        if (!rel && (row < first[region] || row >= last[region])) return -1;
        return producer(region, row);
    };
    struct Val {
        std::vector<uint32_t> uses;   // positions (alive order) of the consumers that can read a slot
        size_t next = 0;
        int slot = -1;
        bool resident = false, force_store = false;
        int dst_slot = -1;                    // the slot it was given when produced
        bool evicted = false;                 // lost its slot before its last use
        uint32_t cell_use_last = 0xffffffffu; // last alive position of an op that reads its *cell* (V_FULL, PICK_INDEX)
    };
    std::vector<Val> vals(2 * (size_t)n_ops);
    std::vector<uint32_t> alive;
    for (uint32_t i = 0; i < n_ops; i++)
        if (!(ops[i].flags & H2E_FLAG_VALUES_SKIP) && kind_of(ops[i]) != K_NONE) alive.push_back(i);
    // pass 1: uses
    std::vector<uint32_t> full_read_last(n_ops, 0);   // per op written "for real" (V_FULL): last alive position reading its rows
    std::vector<uint32_t> alive_pos(n_ops, 0xffffffffu);
    for (uint32_t pos = 0; pos < alive.size(); pos++) alive_pos[alive[pos]] = pos;
    for (uint32_t pos = 0; pos < alive.size(); pos++) {
        const H2EOp& op = ops[alive[pos]];
        int k = kind_of(op);
        for (int q = 0; q < H2E_OP_MAX_REFS; q++) {
            int wtr = writer_of(op.refs[q]);
            if (wtr >= 0 && (uint32_t)wtr != alive[pos] && kind_of(ops[wtr]) == K_FULL) full_read_last[wtr] = std::max(full_read_last[wtr], pos);
        }
        if (k == K_FULL || op.opcode == H2E_OP_PICK_INDEX) {   // reads cells: whatever it reads must be stored
            for (int q = 0; q < H2E_OP_MAX_REFS; q++)
                for (int as_int = 0; as_int < 2; as_int++) {
                    int v = value_of(op.refs[q], as_int != 0);
                    if (v >= 0) {
                        vals[v].force_store = true;
                        vals[v].cell_use_last = pos;
                    }
                }
            continue;
        }
        Opd o[3];
        int n = operands(op, o);
        for (int q = 0; q < n; q++) {
            int v = value_of(o[q].ref, o[q].is_int);
            if (v == -2) throw std::runtime_error("replay compile: operand is not a replay result");
            if (v >= 0) {
                if (ops[v / 2].flags & H2E_FLAG_VALUES_SKIP) throw std::runtime_error("replay compile: live operand of a dead op");
                vals[v].uses.push_back(pos);
            }
        }
    }
    // pass 2: slot allocation
    struct Dec {
        uint8_t kind[3] = {0, 0, 0};
        uint32_t word[3] = {0, 0, 0};
        int val[3] = {-1, -1, -1};
        int dst[2] = {-1, -1};
    };
    std::vector<Dec> dec(alive.size());
    std::vector<int> int_owner(NS, -1), fe_owner(NF, -1);
    auto next_use = [&](int v) -> uint32_t { return vals[v].next < vals[v].uses.size() ? vals[v].uses[vals[v].next] : 0xffffffffu; };
    auto take_slot = [&](std::vector<int>& owner, int v) -> int {
        for (size_t sl = 0; sl < owner.size(); sl++)
            if (owner[sl] < 0) {
                owner[sl] = v;
                return (int)sl;
            }
        size_t victim = 0;
        for (size_t sl = 1; sl < owner.size(); sl++)
            if (next_use(owner[sl]) > next_use(owner[victim])) victim = sl;
        if (next_use(owner[victim]) <= next_use(v)) return -1;   // the new value is the one needed last
        Val& ev = vals[owner[victim]];
        ev.resident = false;
        ev.force_store = true;
        ev.evicted = true;
        ev.slot = -1;
        owner[victim] = v;
        return (int)victim;
    };
    for (uint32_t pos = 0; pos < alive.size(); pos++) {
        uint32_t i = alive[pos];
        const H2EOp& op = ops[i];
        int k = kind_of(op);
        Dec& d = dec[pos];
        if (k != K_FULL && op.opcode != H2E_OP_PICK_INDEX) {
            Opd o[3];
            int n = operands(op, o);
            for (int q = 0; q < n; q++) {
                int v = value_of(o[q].ref, o[q].is_int);
                d.val[q] = v;
                if (v >= 0 && vals[v].resident) {
                    d.kind[q] = o[q].is_int ? H2E_VSRC_INT_SLOT : H2E_VSRC_FE_SLOT;
                    d.word[q] = (uint32_t)vals[v].slot;
                } else {
                    d.kind[q] = H2E_VSRC_GLOBAL;
                    if (v >= 0) {
                        vals[v].force_store = true;
                        vals[v].cell_use_last = pos;   // read through its cell
                    }
                }
            }
            for (int q = 0; q < n; q++) {
                int v = d.val[q];
                if (v < 0) continue;
                if (vals[v].next < vals[v].uses.size() && vals[v].uses[vals[v].next] == pos) vals[v].next++;
            }
            for (int q = 0; q < n; q++) {
                int v = d.val[q];
                if (v < 0 || !vals[v].resident) continue;
                if (vals[v].next >= vals[v].uses.size()) {   // last use: free the slot
                    auto& owner = (kind_of(ops[v / 2]) == K_FE) ? fe_owner : int_owner;
                    owner[vals[v].slot] = -1;
                    vals[v].resident = false;
                }
            }
        }
        int nres = k == K_SEL ? 2 : (k == K_MUL || k == K_ADD || k == K_FE || k == K_CONST) ? 1 : 0;
        for (int w = 0; w < nres; w++) {
            int v = 2 * (int)i + w;
            if (vals[v].uses.empty()) continue;
            int sl = take_slot(k == K_FE ? fe_owner : int_owner, v);
            if (sl < 0) {
                vals[v].force_store = true;
            } else {
                vals[v].slot = sl;
                vals[v].dst_slot = sl;
                vals[v].resident = true;
                d.dst[w] = sl;
            }
        }
    }
    // ---- level-parallel replay ---------------------------------------------------------------------------------
    // A pairing's replay is 175 k ops in one chain, but its dependency graph is only ~8.5 k levels deep (an Fq12
    // product is 54 independent Fq products).  When a segment is that shape, lanes are given to *ops*: a wave replays
    // one instance, each step runs up to 64 independent ops of one opcode, values live in LDS slots shared by the
    // wave (allocated over the step order).  Ops that go through cells (H2E_V_FULL: assign / constants / bisec rows)
    // are steps of their own behind a fence.
    {
        size_t si = (size_t)(sg - r.segments.data());
        bool eligible = !dbg_env("H2E_NO_LEVELS") && sg->n_strands == 1 && alive.size() >= 4096;
        std::vector<uint32_t> level(alive.size(), 0);
        uint32_t depth = 0;
        for (uint32_t pos = 0; pos < alive.size() && eligible; pos++) {
            const H2EOp& op = ops[alive[pos]];
            int k = kind_of(op);
            // (hinted chains are wide, not deep - the MSM tail would need > 1664 value slots - and have pieces instead)
            if (k == K_SEL || op.opcode == H2E_OP_PICK_INDEX || ((op.flags & H2E_FLAG_HINTED) && k == K_MUL)) eligible = false;
            uint32_t lv = 0;
            for (int q = 0; q < 3; q++)
                if (dec[pos].val[q] >= 0) lv = std::max(lv, level[alive_pos[dec[pos].val[q] / 2]] + 1);
            for (int q = 0; q < H2E_OP_MAX_REFS; q++) {
                int wtr = writer_of(op.refs[q]);   // rows an earlier op of this replay writes for real
                if (wtr >= 0 && (uint32_t)wtr != alive[pos] && alive_pos[wtr] != 0xffffffffu && alive_pos[wtr] < pos)
                    lv = std::max(lv, level[alive_pos[wtr]] + 1);
            }
            level[pos] = lv;
            depth = std::max(depth, lv + 1);
        }
        if (eligible && (uint64_t)depth * 4 > alive.size()) eligible = false;   // not shallow enough to pay off
        if (eligible) {
            // as late as possible: an op runs just before its first consumer (values stay in slots for a short time: a
            // pairing's G2 line coefficients are then made next to the Miller-loop step that uses them); ops nothing
            // in the replay depends on run as early as they can, which frees their operands
            std::vector<uint32_t> late(alive.size(), 0xffffffffu);
            for (uint32_t pos = (uint32_t)alive.size(); pos-- > 0;) {
                uint32_t lv = late[pos] == 0xffffffffu ? level[pos] : late[pos] - 1;
                if (lv < level[pos]) throw std::runtime_error("replay compile: level order broken");
                level[pos] = lv;
                const H2EOp& op = ops[alive[pos]];
                for (int q = 0; q < 3; q++)
                    if (dec[pos].val[q] >= 0) {
                        uint32_t pp = alive_pos[dec[pos].val[q] / 2];
                        late[pp] = std::min(late[pp], lv);
                    }
                for (int q = 0; q < H2E_OP_MAX_REFS; q++) {
                    int wtr = writer_of(op.refs[q]);
                    if (wtr >= 0 && (uint32_t)wtr != alive[pos] && alive_pos[wtr] != 0xffffffffu && alive_pos[wtr] < pos)
                        late[alive_pos[wtr]] = std::min(late[alive_pos[wtr]], lv);
                }
            }
        }
        if (eligible) {
            // steps: by level, then by opcode; V_FULL ops one per step
            std::vector<uint32_t> order(alive.size());
            for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
            auto vop_of = [&](uint32_t pos) -> uint32_t {
                const H2EOp& op = ops[alive[pos]];
                if ((op.flags & H2E_FLAG_HINTED) && kind_of(op) == K_MUL) return H2E_V_HINT;
                switch (op.opcode) {
                    case H2E_OP_INT_MUL: return H2E_V_MUL;
                    case H2E_OP_REDUCE: return H2E_V_REDUCE;
                    case H2E_OP_DIV_CORE: return H2E_V_DIV;
                    case H2E_OP_INT_ADD: return H2E_V_ADD;
                    case H2E_OP_INT_SUB: return H2E_V_SUB;
                    case H2E_OP_INT_NEG: return H2E_V_NEG;
                    case H2E_OP_INT_MUL_SMALL: return H2E_V_MUL_SMALL;
                    case H2E_OP_MASK_INT: return H2E_V_MASK;
                    case H2E_OP_BISEC_INT: return H2E_V_BISEC_INT;
                    case H2E_OP_IS_INT_ZERO: return H2E_V_IS_ZERO;
                    case H2E_OP_NOT: return H2E_V_NOT;
                    case H2E_OP_AND: return H2E_V_AND;
                    case H2E_OP_OR: return H2E_V_OR;
                    case H2E_OP_XNOR: return H2E_V_XNOR;
                    case H2E_OP_CONST_INT: return H2E_V_CONST;
                    default: return H2E_V_FULL;
                }
            };
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
                if (level[a] != level[b]) return level[a] < level[b];
                return vop_of(a) < vop_of(b);
            });
            size_t NW = H2E_LEVEL_WAVES;   // steps (waves) per round; 1 in wave mode
            std::vector<uint32_t> step_of(alive.size(), 0);   // the *round* an op runs in (see below)
            std::vector<std::vector<uint32_t>> steps;            // steps[NW * round + wave]: the ops one wave runs in a round
            std::vector<int> lslot;
            int n_slots = 0;
            size_t n_rounds = 0;
            int slot_cap = 0;
            // Two instances per workgroup when their value slots fit side by side in a CU's LDS: a step then holds up to
            // 32 ops, lanes 0-31 run them for one instance and lanes 32-63 for the other (levels are ~20 ops wide on
            // average: one instance leaves two thirds of every wave instruction idle).  Otherwise one instance per
            // workgroup and steps of 64.
            bool paired = false;
            std::function<bool(int)> alloc_slots;
            auto schedule = [&](size_t step_ops, int cap) -> bool {
            steps.clear();
            std::fill(step_of.begin(), step_of.end(), 0u);
            // rounds: H2E_LEVEL_WAVES waves share an instance's value slots; in a round each wave runs one step (up to 64
            // ops of one opcode), all steps of a round come from the same level, a barrier separates rounds.  An op
            // that goes through cells (V_FULL) is a round of its own.
            for (size_t i = 0; i < order.size();) {
                uint32_t lv0 = level[order[i]];
                std::vector<std::vector<uint32_t>> lvl_steps;
                std::vector<uint32_t> lvl_full;
                while (i < order.size() && level[order[i]] == lv0) {
                    uint32_t pos = order[i], vop = vop_of(pos);
                    if (vop == H2E_V_FULL) {
                        lvl_full.push_back(pos);
                        i++;
                        continue;
                    }
                    size_t j = i + 1;
                    while (j < order.size() && j - i < step_ops && level[order[j]] == lv0 && vop_of(order[j]) == vop) j++;
                    lvl_steps.emplace_back(order.begin() + i, order.begin() + j);
                    i = j;
                }
                auto new_round = [&]() {
                    for (size_t w = 0; w < NW; w++) steps.emplace_back();
                    return steps.size() / NW - 1;
                };
                for (uint32_t pos : lvl_full) {
                    size_t rd = new_round();
                    steps[rd * NW].push_back(pos);
                    step_of[pos] = (uint32_t)rd;
                }
                for (size_t k = 0; k < lvl_steps.size(); k++) {
                    if (k % NW == 0) new_round();
                    size_t rd = steps.size() / NW - 1;
                    // Most rounds hold a single step.  Wave w of a workgroup sits on SIMD w of its CU: if that step always
                    // went to wave 0, SIMD 0 would carry the chains of every instance on the CU and the others idle -
                    // the steps rotate over the waves from round to round.
                    steps[rd * NW + (k % NW + rd) % NW] = lvl_steps[k];
                    for (uint32_t pos : lvl_steps[k]) step_of[pos] = (uint32_t)rd;
                }
            }
            return alloc_slots(cap);
            };   // schedule
            // ---- cost-class rounds (default) ---------------------------------------------------------------------
            // Rounds by level put a product into 62 % of the rounds of a pairing check although its multiplicative depth
            // is a tenth of its depth: an Fq12 product is one level of int_mul between a dozen levels of int_add / int_sub /
            // reduce, and ALAP levels scatter the products of independent branches over all of them.  A round costs what
            // its most expensive lane costs (a product ~2.5 us, a reduce ~1 us, an addition ~0.5 us), so the rounds are
            // built by cost class instead: walking the dependency graph from the results backwards (as late as possible:
            // a value is made just before its first use, which keeps the value slots few), a round takes every ready op of
            // the *cheapest* class that has ready ops - expensive ops wait until nothing cheaper can go, and so meet the
            // products of the other branches in one round.  Light ops of different opcodes share a step (the kernel
            // dispatches them per lane: H2E_VFLAG_MIXED); products, reduces and divisions keep one opcode per step.
            auto cls_of = [&](uint32_t pos) -> int {   // 0 light, 1 medium, 2 heavy, 3 through cells
                switch (vop_of(pos)) {
                    case H2E_V_FULL: return 3;
                    case H2E_V_MUL: case H2E_V_DIV: return 2;
                    case H2E_V_REDUCE: case H2E_V_MUL_SMALL: case H2E_V_CONST: case H2E_V_HINT: return 1;
                    default: return 0;
                }
            };
            std::vector<std::vector<uint32_t>> preds(alive.size()), succs(alive.size());
            auto build_deps = [&]() {
                if (!succs.empty() && !preds.empty() && (!preds[alive.size() - 1].empty() || !succs[0].empty())) return;
                for (uint32_t pos = 0; pos < alive.size(); pos++) {
                    const H2EOp& op = ops[alive[pos]];
                    auto add = [&](uint32_t pp) {
                        if (pp == pos || pp == 0xffffffffu) return;
                        if (std::find(preds[pos].begin(), preds[pos].end(), pp) != preds[pos].end()) return;
                        preds[pos].push_back(pp);
                        succs[pp].push_back(pos);
                    };
                    for (int q = 0; q < 3; q++)
                        if (dec[pos].val[q] >= 0) add(alive_pos[dec[pos].val[q] / 2]);
                    for (int q = 0; q < H2E_OP_MAX_REFS; q++) {
                        int wtr = writer_of(op.refs[q]);
                        if (wtr >= 0 && (uint32_t)wtr != alive[pos] && alive_pos[wtr] != 0xffffffffu && alive_pos[wtr] < pos) add(alive_pos[wtr]);
                    }
                }
            };
            double modelled_us = 0;
            auto schedule_classes = [&](size_t step_ops, int cap, int policy) -> bool {
                build_deps();
                steps.clear();
                std::fill(step_of.begin(), step_of.end(), 0u);
                const size_t n = alive.size();
                // sinks (results nothing in the replay reads: they are only stored) do not take part in the backward walk:
                // at the end of the program they would pin their operands' slots; they are placed forward afterwards
                std::vector<uint8_t> is_sink(n, 0);
                for (uint32_t pos = 0; pos < n; pos++) is_sink[pos] = succs[pos].empty() ? 1 : 0;
                std::vector<uint32_t> left(n, 0);
                for (uint32_t pos = 0; pos < n; pos++)
                    for (uint32_t s : succs[pos])
                        if (!is_sink[s]) left[pos]++;
                // ready ops by class; an op released while round r is formed may go into round r - 1 (backwards) at the earliest
                std::vector<uint32_t> ready[4], released;
                std::vector<uint8_t> done(n, 0);
                size_t n_left = 0;
                for (uint32_t pos = 0; pos < n; pos++) {
                    if (is_sink[pos]) {   // a sink whose operands are sinks' operands only: its preds count it as scheduled
                        continue;
                    }
                    n_left++;
                    if (left[pos] == 0) ready[cls_of(pos)].push_back(pos);
                }
                // sinks with non-sink consumers do not exist; sinks release their preds right away
                std::vector<std::vector<std::vector<uint32_t>>> rounds_rev;   // [round][step] -> ops
                std::vector<int> round_cls_rev;
                auto take_steps = [&](std::vector<uint32_t>& pool, bool uniform_vop, std::vector<std::vector<uint32_t>>& rsteps) {
                    // as many steps as the round has waves left; ops that do not fit stay in the pool
                    std::vector<uint32_t> rest;
                    if (!uniform_vop) {
                        size_t i = 0;
                        while (i < pool.size() && rsteps.size() < NW) {
                            size_t j = std::min(pool.size(), i + step_ops);
                            rsteps.emplace_back(pool.begin() + i, pool.begin() + j);
                            i = j;
                        }
                        rest.assign(pool.begin() + i, pool.end());
                    } else {
                        std::stable_sort(pool.begin(), pool.end(), [&](uint32_t a, uint32_t b) { return vop_of(a) < vop_of(b); });
                        size_t i = 0;
                        while (i < pool.size()) {
                            size_t j = i + 1;
                            while (j < pool.size() && j - i < step_ops && vop_of(pool[j]) == vop_of(pool[i])) j++;
                            if (rsteps.size() < NW) rsteps.emplace_back(pool.begin() + i, pool.begin() + j);
                            else rest.insert(rest.end(), pool.begin() + i, pool.begin() + j);
                            i = j;
                        }
                    }
                    pool.swap(rest);
                };
                while (n_left > 0) {
                    std::vector<std::vector<uint32_t>> rsteps;
                    int rc = -1;
                    if (!ready[0].empty()) {
                        rc = 0;
                        take_steps(ready[0], false, rsteps);
                    } else if (!ready[3].empty()) {
                        rc = 3;
                        rsteps.push_back({ready[3].back()});
                        ready[3].pop_back();
                    } else if (policy == 0 ? !ready[1].empty() : (ready[2].empty() && !ready[1].empty())) {
                        rc = 1;   // policy 0: cheapest class first; policy 1: reduces ride along with products when both are ready
                        take_steps(ready[1], true, rsteps);
                    } else if (!ready[2].empty()) {
                        rc = 2;
                        take_steps(ready[2], true, rsteps);
                        if (policy == 1 && rsteps.size() < NW) take_steps(ready[1], true, rsteps);
                    } else {
                        throw std::runtime_error("replay compile: class scheduler stalled");
                    }
                    for (auto& st : rsteps)
                        for (uint32_t pos : st) {
                            done[pos] = 1;
                            n_left--;
                            for (uint32_t pp : preds[pos])
                                if (--left[pp] == 0) released.push_back(pp);
                        }
                    for (uint32_t pp : released) ready[cls_of(pp)].push_back(pp);
                    released.clear();
                    rounds_rev.push_back(std::move(rsteps));
                    round_cls_rev.push_back(rc);
                }
                // forward order
                std::vector<std::vector<std::vector<uint32_t>>> rounds(rounds_rev.rbegin(), rounds_rev.rend());
                std::vector<int> round_cls(round_cls_rev.rbegin(), round_cls_rev.rend());
                std::vector<uint32_t> round_of(n, 0);
                for (size_t rd = 0; rd < rounds.size(); rd++)
                    for (auto& st : rounds[rd])
                        for (uint32_t pos : st) round_of[pos] = (uint32_t)rd;
                // sinks: the first round after their operands that can take them without getting more expensive
                for (uint32_t pos = 0; pos < n; pos++) {
                    if (!is_sink[pos]) continue;
                    size_t rd0 = 0;
                    for (uint32_t pp : preds[pos]) {
                        if (is_sink[pp] && !done[pp]) throw std::runtime_error("replay compile: sink reads an unplaced sink");
                        rd0 = std::max<size_t>(rd0, (size_t)round_of[pp] + 1);
                    }
                    int c = cls_of(pos);
                    uint32_t vop = vop_of(pos);
                    bool placed = false;
                    for (size_t rd = rd0; rd < rounds.size() && !placed; rd++) {
                        if (round_cls[rd] == 3 || c == 3) continue;
                        if (round_cls[rd] < c) continue;   // would make the round more expensive
                        for (auto& st : rounds[rd]) {
                            bool light_step = cls_of(st[0]) == 0;
                            if (st.size() < step_ops && ((c == 0 && light_step) || (c != 0 && vop_of(st[0]) == vop))) {
                                st.push_back(pos);
                                placed = true;
                                break;
                            }
                        }
                        if (!placed && rounds[rd].size() < NW) {
                            rounds[rd].push_back({pos});
                            placed = true;
                        }
                        if (placed) round_of[pos] = (uint32_t)rd;
                    }
                    if (!placed) {
                        rounds.push_back({{pos}});
                        round_cls.push_back(c);
                        round_of[pos] = (uint32_t)rounds.size() - 1;
                    }
                    done[pos] = 1;
                }
                size_t cnt[4] = {0, 0, 0, 0};
                for (size_t rd = 0; rd < rounds.size(); rd++) {
                    for (size_t w = 0; w < NW; w++) steps.emplace_back();
                    // heavier steps first, rotated over the waves like the level rounds
                    for (size_t k = 0; k < rounds[rd].size(); k++) {
                        steps[rd * NW + (round_cls[rd] == 3 ? k : (k % NW + rd) % NW)] = rounds[rd][k];
                        for (uint32_t pos : rounds[rd][k]) step_of[pos] = (uint32_t)rd;
                    }
                    cnt[round_cls[rd]]++;
                }
                n_rounds = rounds.size();
                modelled_us = 0.55 * cnt[0] + 1.2 * cnt[1] + 3.0 * cnt[2] + 3.0 * cnt[3];
                if (dbg_env("H2E_DUMP_TAPE"))
                    fprintf(stderr, "   class rounds (policy %d, steps of %zu): %zu light, %zu medium, %zu heavy, %zu through cells = %zu rounds, modelled %.2f ms\n",
                            policy, step_ops, cnt[0], cnt[1], cnt[2], cnt[3], n_rounds, modelled_us * 1e-3);
                return alloc_slots(cap);
            };
            alloc_slots = [&](int cap) -> bool {
            slot_cap = cap;
            n_rounds = steps.size() / NW;
            // value slots over the round order: a slot freed in round r is reusable from round r + 1
            std::vector<uint32_t> last_step(2 * (size_t)n_ops, 0);
            for (uint32_t pos = 0; pos < alive.size(); pos++)
                for (int q = 0; q < 3; q++)
                    if (dec[pos].val[q] >= 0) last_step[dec[pos].val[q]] = std::max(last_step[dec[pos].val[q]], step_of[pos]);
            lslot.assign(2 * (size_t)n_ops, -1);
            std::vector<std::vector<int>> free_at(n_rounds + 1);
            std::vector<int> free_list;
            n_slots = 0;
            bool fits = true;
            for (size_t rd = 0; rd < n_rounds && fits; rd++) {
                for (int sl : free_at[rd]) free_list.push_back(sl);
                for (size_t w = 0; w < NW; w++)
                    for (uint32_t pos : steps[rd * NW + w]) {
                        uint32_t i = alive[pos];
                        int k = kind_of(ops[i]);
                        int nres = (k == K_MUL || k == K_ADD || k == K_FE || k == K_CONST) ? 1 : 0;
                        for (int ww = 0; ww < nres; ww++) {
                            int v = 2 * (int)i + ww;
                            if (vals[v].uses.empty()) continue;
                            int sl;
                            if (!free_list.empty()) {
                                sl = free_list.back();
                                free_list.pop_back();
                            } else {
                                sl = n_slots++;
                            }
                            lslot[v] = sl;
                            free_at[std::min<size_t>(last_step[v] + 1, n_rounds)].push_back(sl);
                        }
                    }
                if (n_slots > slot_cap) fits = false;
            }
            return fits;
            };   // schedule
            // A/B knobs of the program compiler, read when a program is recorded (never while a run is queued):
            // H2E_LEVEL_MODE=pair (default) | single | wave : kernel shape - four waves and two / one instance(s) per workgroup
            //     (h2e_replay_levels) | one wave per instance, no barriers, compact records streamed through LDS (h2e_replay_wave:
            //     measured 26.5 vs 24.3 ms for 64 bn256 checks - a light round is ~3 k cycles of multi-word additions either way);
            // H2E_LEVEL_SCHED=levels|classes0|classes1 (default: the cheaper of the two class policies by the cost model;
            //     `levels` only with the four-wave kernels)
            bool by_classes = true, wave_mode = false;
            {
                const uint32_t slot_bytes = (2 * (uint32_t)L + 4) * 8;
                const int cap_pair = (int)((160u * 1024 - 4u * 1024) / 2 / slot_bytes), cap_single = (int)((160u * 1024 - 30u * 1024) / slot_bytes);
                const char* mode = getenv("H2E_LEVEL_SCHED");
                const char* kmode = dbg_env("H2E_LEVEL_MODE");   // ("wave": a kernel of -DH2E_AB_KERNELS engine units only)
                const bool allow_pair = !(kmode && !strcmp(kmode, "single"));
                wave_mode = kmode && !strcmp(kmode, "wave") && !(mode && !strcmp(mode, "levels"));
                int forced = mode && !strcmp(mode, "classes0") ? 0 : mode && !strcmp(mode, "classes1") ? 1 : -1;
                auto best_policy = [&](size_t step_ops, int cap) -> bool {
                    double best = 0;
                    int pick = -1;
                    for (int pol = 0; pol < 2; pol++) {
                        if (forced >= 0 && pol != forced) continue;
                        if (NW == 1 && pol == 1) continue;   // one step per round: nothing can ride along
                        if (schedule_classes(step_ops, cap, pol) && (pick < 0 || modelled_us < best)) {
                            pick = pol;
                            best = modelled_us;
                        }
                    }
                    if (pick < 0) return false;
                    return schedule_classes(step_ops, cap, pick);
                };
                if (wave_mode) {
                    NW = 1;
                    // (chunk buffers, ceil tables and a margin for other workgroups' static LDS come off the CU's 160 KB)
                    eligible = best_policy(64, (int)((160u * 1024 - 2u * H2E_WCHUNK * 32u - 8u * 1024) / slot_bytes));
                } else if (mode && !strcmp(mode, "levels")) {
                    by_classes = false;
                    paired = allow_pair && schedule(32, cap_pair);
                    if (!paired) eligible = schedule(64, cap_single);
                } else {
                    paired = allow_pair && best_policy(32, cap_pair);
                    if (!paired) eligible = best_policy(64, cap_single);
                }
            }
            if (!eligible && dbg_env("H2E_DUMP_TAPE"))
                fprintf(stderr, "segment %zu: level-parallel replay needs more than %d value slots (depth %u, %zu rounds)\n", si, slot_cap, depth, n_rounds);
            // one op of the schedule as a level record
            auto make_rec = [&](uint32_t pos, bool mixed) -> H2EVRec {
                H2EVRec h{{0, 0, 0, 0, 0, 0, 0, 0}};
                uint32_t i = alive[pos];
                const H2EOp& op = ops[i];
                int k = kind_of(op);
                uint32_t vop = vop_of(pos), vflags = 0;
                bool store = !(op.flags & H2E_FLAG_LOCAL_RESULT) || vals[2 * (size_t)i].force_store;
                if (op.opcode == H2E_OP_AND || op.opcode == H2E_OP_OR || op.opcode == H2E_OP_XNOR || op.opcode == H2E_OP_BISEC_INT)
                    store = true;
                if (store) vflags |= H2E_VFLAG_STORE;
                if (vop == H2E_V_HINT && (op.flags & H2E_FLAG_HINT_STRIDED)) vflags |= H2E_VFLAG_HINT_STRIDED;
                if (mixed) vflags |= H2E_VFLAG_MIXED;
                int dsl = lslot[2 * (size_t)i];
                h.w[0] = vop | (vflags << 8) | ((uint32_t)(dsl >= 0 ? dsl : 0xffff) << 16);
                h.w[1] = vop == H2E_V_FULL ? i : op.imm;   // V_FULL: index of the tape op (segment relative)
                h.w[5] = k == K_FE ? fe_row(op) : op.base_row;
                h.w[6] = op.range_row;
                if (vop != H2E_V_FULL) {
                    Opd o[3];
                    int n = operands(op, o);
                    for (int q = 0; q < n; q++) {
                        int v = dec[pos].val[q];
                        if (v >= 0) {
                            h.w[7] |= (uint32_t)(o[q].is_int ? H2E_VSRC_INT_SLOT : H2E_VSRC_FE_SLOT) << (3 * q);
                            h.w[2 + q] = (uint32_t)lslot[v];
                        } else {
                            h.w[7] |= (uint32_t)H2E_VSRC_GLOBAL << (3 * q);
                            if (o[q].is_int) {
                                h.w[2 + q] = (uint32_t)h_lrefs.size();
                                for (int j = 0; j <= L; j++) h_lrefs.push_back(op.refs[o[q].refpos + j]);
                            } else {
                                h.w[2 + q] = o[q].ref;
                            }
                        }
                    }
                }
                return h;
            };
            if (eligible && wave_mode) {
                // compact records in round order, padded so that no round straddles an H2E_WCHUNK-record chunk (the kernel
                // streams the records through two LDS chunk buffers); per round: first record, count | kind << 8
                // (kind: 0 = light ops of mixed opcodes, else the round's one opcode)
                while (h_lrecs.size() % H2E_WCHUNK) h_lrecs.push_back(H2EVRec{{H2E_V_NOP, 0, 0, 0, 0, 0, 0, 0}});
                seg_l_begin[si] = (uint32_t)h_lrecs.size();
                seg_lr_begin[si] = (uint32_t)h_lrounds.size();
                seg_l_steps[si] = (uint32_t)n_rounds;
                seg_l_slots[si] = (uint32_t)std::max(1, n_slots);
                seg_l_pair[si] = 2u;
                for (size_t rd = 0; rd < n_rounds; rd++) {
                    auto& stp = steps[rd];
                    if (stp.empty() || stp.size() > 64) throw std::runtime_error("replay compile: bad wave round");
                    size_t at = h_lrecs.size() - seg_l_begin[si];
                    if (at % H2E_WCHUNK + stp.size() > H2E_WCHUNK)
                        while ((h_lrecs.size() - seg_l_begin[si]) % H2E_WCHUNK) h_lrecs.push_back(H2EVRec{{H2E_V_NOP, 0, 0, 0, 0, 0, 0, 0}});
                    at = h_lrecs.size() - seg_l_begin[si];
                    const bool mixed = cls_of(stp[0]) == 0;
                    h_lrounds.push_back((uint32_t)at);
                    h_lrounds.push_back((uint32_t)stp.size() | ((mixed ? 0u : vop_of(stp[0])) << 8));
                    for (uint32_t pos : stp) h_lrecs.push_back(make_rec(pos, mixed));
                }
                while ((h_lrecs.size() - seg_l_begin[si]) % H2E_WCHUNK) h_lrecs.push_back(H2EVRec{{H2E_V_NOP, 0, 0, 0, 0, 0, 0, 0}});
                seg_l_recs[si] = (uint32_t)(h_lrecs.size() - seg_l_begin[si]);
            }
            if (eligible && !wave_mode) {
                seg_l_begin[si] = (uint32_t)h_lrecs.size();
                seg_l_steps[si] = (uint32_t)n_rounds;
                seg_l_slots[si] = (uint32_t)std::max(1, n_slots);
                seg_l_pair[si] = paired ? 1u : 0u;
                for (size_t sidx = 0; sidx < steps.size(); sidx++) {
                    auto& stp = steps[sidx];
                    // the other waves of a V_FULL round fence their stores before the barrier (lane 0 of their NOP step says so)
                    bool full_round = false;
                    for (size_t w = 0; w < NW; w++) {
                        const auto& other = steps[sidx / NW * NW + w];
                        full_round = full_round || (!other.empty() && vop_of(other[0]) == H2E_V_FULL);
                    }
                    const size_t step_lanes = paired ? 32 : 64;
                    // a step of light ops holds any mix of their opcodes (class rounds): the kernel dispatches per lane
                    const bool mixed = by_classes && !stp.empty() && cls_of(stp[0]) == 0;
                    std::vector<H2EVRec> step_recs;
                    for (size_t lane = 0; lane < step_lanes; lane++) {
                        H2EVRec h{{H2E_V_NOP | ((full_round && lane == 0) ? (H2E_VFLAG_FENCE << 8) : 0u), 0, 0, 0, 0, 0, 0, 0}};
                        if (lane < stp.size()) h = make_rec(stp[lane], mixed);
                        step_recs.push_back(h);
                    }
                    // (paired: the second half of the wave runs the same records for the workgroup's other instance)
                    for (size_t rep2 = 0; rep2 < 64 / step_lanes; rep2++) h_lrecs.insert(h_lrecs.end(), step_recs.begin(), step_recs.end());
                }
            }
            if (eligible) {
                if (dbg_env("H2E_DUMP_TAPE")) {
                    fprintf(stderr, "segment %zu: level-parallel replay: %zu alive ops, depth %u, %zu rounds of %zu waves, %d value slots, %s\n", si,
                            alive.size(), depth, n_rounds, NW, n_slots, wave_mode ? "one wave per instance" : paired ? "two instances per workgroup" : "one instance per workgroup");
                    // rounds by their most expensive op kind, and how many of them read an operand from global cells
                    std::map<uint32_t, std::pair<size_t, size_t>> by_vop;
                    size_t global_rounds = 0, global_operands = 0;
                    for (size_t rd = 0; rd < n_rounds; rd++) {
                        uint32_t worst = 0;
                        bool g = false;
                        for (size_t w = 0; w < NW; w++)
                            for (uint32_t pos : steps[rd * NW + w]) {
                                uint32_t vop = vop_of(pos);
                                auto rank = [](uint32_t v) { return v == H2E_V_FULL ? 100u : v == H2E_V_DIV ? 90u : v == H2E_V_MUL ? 80u : v == H2E_V_REDUCE ? 70u : 10u; };
                                if (rank(vop) > rank(worst) || worst == 0) worst = vop;
                                Opd o[3];
                                int n = vop == H2E_V_FULL ? 0 : operands(ops[alive[pos]], o);
                                for (int q = 0; q < n; q++)
                                    if (dec[pos].val[q] < 0) {
                                        g = true;
                                        global_operands++;
                                    }
                            }
                        by_vop[worst].first++;
                        if (g) {
                            by_vop[worst].second++;
                            global_rounds++;
                        }
                    }
                    for (auto& kv : by_vop) fprintf(stderr, "   rounds led by vop %u: %zu (%zu with a global operand)\n", kv.first, kv.second.first, kv.second.second);
                    fprintf(stderr, "   %zu rounds with global operands, %zu global operands in all\n", global_rounds, global_operands);
                }
            }
        }
    }
    // ---- pieces ------------------------------------------------------------------------------------------
    // The replay is one dependent chain only through values.  Where every value that is live across a position
    // can be rebuilt from hints / external cells by a few ops (its producers' closure), the chain is cut there:
    // the next piece starts with that closure as a prologue (results not stored) and runs in its own lanes.
    std::vector<uint32_t> pos_of_op(n_ops, 0xffffffffu);
    for (uint32_t pos = 0; pos < alive.size(); pos++) pos_of_op[alive[pos]] = pos;
    const uint32_t INF = 0xffffffffu;
    auto prod_pos = [&](int v) { return pos_of_op[v / 2]; };
    auto last_use = [&](int v) -> uint32_t { return vals[v].uses.empty() ? 0 : vals[v].uses.back(); };
    // values produced by the replay in program order, for the live-set scan
    std::vector<int> produced;
    for (uint32_t pos = 0; pos < alive.size(); pos++) {
        int k = kind_of(ops[alive[pos]]);
        int nres = k == K_SEL ? 2 : (k == K_MUL || k == K_ADD || k == K_FE || k == K_CONST) ? 1 : 0;
        for (int w = 0; w < nres; w++) produced.push_back(2 * (int)alive[pos] + w);
    }
    auto is_fe_val = [&](int v) { return kind_of(ops[v / 2]) == K_FE; };
    struct Restart {
        uint32_t pos;
        std::vector<uint32_t> prologue;            // alive positions, program order
        std::map<int, int> slot_of;                // value -> slot during the prologue
    };
    const uint32_t PIECE_TARGET = 96, PIECE_BUDGET = 40;
    static const bool pieces_on = !dbg_env("H2E_NO_PIECES");
    static const bool stage_on = !dbg_env("H2E_NO_STAGE");
    // positions that can never be cut: an op at or after p reads a *cell* written before p (rows of a V_FULL op, or
    // a value that lost / never had its slot) - difference arrays over (writer, last reader]
    std::vector<int32_t> blocked(alive.size() + 2, 0);
    for (uint32_t q = 0; q < alive.size(); q++)
        if (kind_of(ops[alive[q]]) == K_FULL && full_read_last[alive[q]] > q) {
            blocked[q + 1]++;
            blocked[full_read_last[alive[q]] + 1]--;
        }
    for (int v : produced)
        if (vals[v].cell_use_last != INF && vals[v].cell_use_last > prod_pos(v)) {
            blocked[prod_pos(v) + 1]++;
            blocked[vals[v].cell_use_last + 1]--;
        }
    for (size_t q = 1; q < blocked.size(); q++) blocked[q] += blocked[q - 1];
    // values whose last slot-use is at a given position (to keep the live set incrementally)
    std::vector<std::vector<int>> expires(alive.size() + 1);
    for (int v : produced)
        if (!vals[v].uses.empty()) expires[last_use(v)].push_back(v);
    auto try_restart = [&](uint32_t p, const std::set<int>& live_set, Restart& rs) -> bool {
        if (blocked[p] > 0) return false;
        std::vector<int> live;
        for (int v : live_set) {
            if (vals[v].dst_slot < 0 || vals[v].evicted) return false;
            live.push_back(v);
        }
        std::set<uint32_t> closure;
        std::set<int> cvals;
        std::vector<int> work(live.begin(), live.end());
        while (!work.empty()) {
            int v = work.back();
            work.pop_back();
            uint32_t q = prod_pos(v);
            // (both results of a SELECT_POINT come from one op)
            cvals.insert(v);
            if (!closure.insert(q).second) continue;
            if (closure.size() > PIECE_BUDGET) return false;
            if (kind_of(ops[alive[q]]) == K_FULL) return false;
            for (int j = 0; j < 3; j++)
                if (dec[q].val[j] >= 0) work.push_back(dec[q].val[j]);
        }
        rs.pos = p;
        rs.prologue.assign(closure.begin(), closure.end());
        std::vector<bool> int_used(NS, false), fe_used(NF, false);
        for (int v : live) {
            rs.slot_of[v] = vals[v].dst_slot;
            (is_fe_val(v) ? fe_used : int_used)[vals[v].dst_slot] = true;
        }
        for (int v : cvals) {
            if (rs.slot_of.count(v)) continue;
            auto& used = is_fe_val(v) ? fe_used : int_used;
            int sl = -1;
            for (size_t t = 0; t < used.size(); t++)
                if (!used[t]) {
                    sl = (int)t;
                    break;
                }
            if (sl < 0) return false;
            used[sl] = true;
            rs.slot_of[v] = sl;
        }
        return true;
    };
    std::vector<Restart> restarts;
    const uint32_t INT_UNITS = (uint32_t)L + 2, HINT_UNITS = (uint32_t)r.fp.w_words / 2, FE_UNITS = 2;
    const uint32_t UNIT_TARGET = 44, UNIT_MAX = 64;   // staging units (16 bytes per lane) a piece may gather
    auto stageable = [&](uint32_t ref) { return ref != H2E_NO_REF && H2E_REF_REGION(ref) != H2E_REGION_PARAM && writer_of(ref) < 0; };
    auto units_of = [&](uint32_t pos) -> uint32_t {   // upper estimate of the memory inputs of one op
        const H2EOp& op = ops[alive[pos]];
        if ((op.flags & H2E_FLAG_HINTED) && kind_of(op) == K_MUL) return HINT_UNITS;
        if (op.opcode == H2E_OP_SELECT_POINT && (op.flags & H2E_FLAG_PRESELECTED)) return 2 * HINT_UNITS;
        Opd o[3];
        int n = operands(op, o);
        uint32_t u = 0;
        for (int q = 0; q < n; q++)
            if (dec[pos].val[q] < 0 && stageable(o[q].ref)) u += o[q].is_int ? INT_UNITS : FE_UNITS;
        return u;
    };
    if (pieces_on) {
        uint32_t since = 0, units = 0, retry_at = 0;
        std::set<int> live_set;   // values produced before pos with a slot-use at or after pos
        for (uint32_t pos = 1; pos < alive.size(); pos++) {
            {   // advance the live set from pos - 1 to pos
                uint32_t i = alive[pos - 1];
                int k = kind_of(ops[i]);
                int nres = k == K_SEL ? 2 : (k == K_MUL || k == K_ADD || k == K_FE || k == K_CONST) ? 1 : 0;
                for (int w = 0; w < nres; w++)
                    if (!vals[2 * (size_t)i + w].uses.empty() && last_use(2 * (int)i + w) >= pos) live_set.insert(2 * (int)i + w);
                for (int v : expires[pos - 1]) live_set.erase(v);
            }
            since++;
            units += units_of(pos - 1);
            if ((since < PIECE_TARGET && units < UNIT_TARGET) || pos < retry_at) continue;
            Restart rs;
            if (try_restart(pos, live_set, rs)) {
                restarts.push_back(std::move(rs));
                since = 0;
                units = 0;
            } else {
                retry_at = pos + 4;   // (a failed attempt costs a closure walk: do not try every position)
            }
        }
    }
    // ---- emit ------------------------------------------------------------------------------------------------
    std::vector<H2EVRec> out;
    auto pad_chunk = [&]() {
        while (out.size() % H2E_VCHUNK) out.push_back(H2EVRec{{H2E_V_NOP, 0, 0, 0, 0, 0, 0, 0}});
    };
    auto pad_to = [&](size_t need) {
        if (out.size() % H2E_VCHUNK + need > H2E_VCHUNK) pad_chunk();
    };
    // remap: nullptr = the op in its own place; else the prologue copy (slots from the map, nothing stored)
    struct StageMap {
        std::map<uint32_t, uint32_t> hint, cells, sel;   // hint slot | strided << 31 -> unit ; first cell ref -> unit ; selection entry -> unit
        uint32_t units = 0;
    };
    auto emit = [&](uint32_t pos, const std::map<int, int>* remap, const StageMap& sm) {
        uint32_t i = alive[pos];
        const H2EOp& op = ops[i];
        int k = kind_of(op);
        const Dec& d = dec[pos];
        std::vector<uint32_t> ext;
        H2EVRec h{{0, 0, 0, 0, 0, 0, 0, 0}};
        auto dst_of = [&](int w) -> uint32_t {
            if (!remap) return d.dst[w] >= 0 ? (uint32_t)d.dst[w] : H2E_V_NO_SLOT;
            auto it = remap->find(2 * (int)i + w);
            return it == remap->end() ? H2E_V_NO_SLOT : (uint32_t)it->second;
        };
        uint32_t vop = H2E_V_NOP, vflags = 0, dst = dst_of(0);
        bool hinted = (op.flags & H2E_FLAG_HINTED) != 0;
        switch (op.opcode) {
            case H2E_OP_INT_MUL: vop = hinted ? H2E_V_HINT : H2E_V_MUL; break;
            case H2E_OP_REDUCE: vop = hinted ? H2E_V_HINT : H2E_V_REDUCE; break;
            case H2E_OP_DIV_CORE: vop = hinted ? H2E_V_HINT : H2E_V_DIV; break;
            case H2E_OP_INT_ADD: vop = H2E_V_ADD; break;
            case H2E_OP_INT_SUB: vop = H2E_V_SUB; break;
            case H2E_OP_INT_NEG: vop = H2E_V_NEG; break;
            case H2E_OP_INT_MUL_SMALL: vop = H2E_V_MUL_SMALL; break;
            case H2E_OP_MASK_INT: vop = H2E_V_MASK; break;
            case H2E_OP_BISEC_INT: vop = H2E_V_BISEC_INT; break;
            case H2E_OP_IS_INT_ZERO: vop = H2E_V_IS_ZERO; break;
            case H2E_OP_NOT: vop = H2E_V_NOT; break;
            case H2E_OP_AND: vop = H2E_V_AND; break;
            case H2E_OP_OR: vop = H2E_V_OR; break;
            case H2E_OP_XNOR: vop = H2E_V_XNOR; break;
            case H2E_OP_PICK_INDEX: vop = H2E_V_PICK_INDEX; break;
            case H2E_OP_SELECT_POINT: vop = (op.flags & H2E_FLAG_PRESELECTED) ? H2E_V_LOAD_SEL : H2E_V_SELECT_POINT; break;
            case H2E_OP_CONST_INT: vop = H2E_V_CONST; break;
            default: vop = H2E_V_FULL; break;
        }
        if (hinted && (op.flags & H2E_FLAG_HINT_STRIDED)) vflags |= H2E_VFLAG_HINT_STRIDED;
        uint32_t imm = op.imm;
        if (vop == H2E_V_HINT) {
            auto it = sm.hint.find(op.imm | ((op.flags & H2E_FLAG_HINT_STRIDED) ? 0x80000000u : 0));
            if (it != sm.hint.end()) {
                vflags |= H2E_VFLAG_STAGED;
                imm = it->second;
            }
        }
        bool store = !(op.flags & H2E_FLAG_LOCAL_RESULT) || vals[2 * (size_t)i].force_store || vals[2 * (size_t)i + 1].force_store;
        if (k == K_SEL || op.opcode == H2E_OP_AND || op.opcode == H2E_OP_OR || op.opcode == H2E_OP_XNOR || op.opcode == H2E_OP_PICK_INDEX ||
            op.opcode == H2E_OP_BISEC_INT)
            store = true;   // never flagged local
        if (vop == H2E_V_LOAD_SEL) store = false;   // the expansion writes the select rows
        if (store && !remap) vflags |= H2E_VFLAG_STORE;
        h.w[1] = imm;
        h.w[5] = k == K_FE ? fe_row(op) : op.base_row;
        h.w[6] = k == K_SEL ? op.select_row : op.range_row;
        if (vop == H2E_V_FULL) {
            const uint32_t* raw = (const uint32_t*)&op;
            ext.assign(raw, raw + 16);
        } else if (vop == H2E_V_PICK_INDEX) {
            for (uint32_t q = 0; q < op.imm && q < 5; q++) ext.push_back(op.refs[q]);
        } else if (vop == H2E_V_LOAD_SEL) {
            auto it = sm.sel.find(op.refs[1]);
            if (it != sm.sel.end()) {
                h.w[2] = it->second;
                h.w[7] |= H2E_VSRC_STAGE;
            } else {
                h.w[2] = op.refs[1];
                h.w[7] |= H2E_VSRC_GLOBAL;
            }
            h.w[7] |= dst_of(1) << 16;
        } else {
            Opd o[3];
            int n = operands(op, o);
            for (int q = 0; q < n; q++) {
                uint32_t kind = d.kind[q], word = d.word[q];
                if (remap && d.val[q] >= 0) {   // a value of the closure: in the slot the prologue gave it
                    kind = o[q].is_int ? H2E_VSRC_INT_SLOT : H2E_VSRC_FE_SLOT;
                    word = (uint32_t)remap->at(d.val[q]);
                }
                if (kind == H2E_VSRC_GLOBAL && d.val[q] < 0) {   // an input from memory: staged by this piece's gather?
                    auto it = sm.cells.find(o[q].ref);
                    if (it != sm.cells.end()) {
                        kind = H2E_VSRC_STAGE;
                        word = it->second;
                    }
                }
                h.w[7] |= kind << (3 * q);
                if (kind != H2E_VSRC_GLOBAL) {
                    h.w[2 + q] = word;
                } else if (o[q].is_int) {
                    h.w[2 + q] = (uint32_t)ext.size();
                    for (int j = 0; j <= L; j++) ext.push_back(op.refs[o[q].refpos + j]);
                } else {
                    h.w[2 + q] = o[q].ref;
                }
            }
            if (k == K_SEL) h.w[7] |= dst_of(1) << 16;
        }
        uint32_t n_ext = (uint32_t)((ext.size() + 7) / 8);
        h.w[0] = vop | (vflags << 8) | (dst << 16) | (n_ext << 24);
        pad_to(1 + n_ext);
        out.push_back(h);
        ext.resize((size_t)n_ext * 8, H2E_NO_REF);
        for (uint32_t e = 0; e < n_ext; e++) {
            H2EVRec x;
            for (int j = 0; j < 8; j++) x.w[j] = ext[e * 8 + j];
            out.push_back(x);
        }
    };
    size_t si = (size_t)(sg - r.segments.data());
    uint32_t vbase = (uint32_t)h_vtape.size();   // multiple of H2E_VCHUNK
    seg_piece_begin[si] = (uint32_t)h_vpieces.size() / 2;
    uint32_t max_units = 0;
    // LDS sizing: integer slots actually used; what is left of 130 KB (28 KB are static: row staging for H2E_V_FULL ops,
    // record chunks; h2e_engine_launch re-checks the sum) bounds the staging units
    uint32_t used_slots = 1;
    for (auto& d : dec)
        for (int w = 0; w < 2; w++)
            if (d.dst[w] >= 0 && kind_of(ops[alive[&d - dec.data()]]) != K_FE) used_slots = std::max(used_slots, (uint32_t)d.dst[w] + 1);
    for (auto& rs : restarts)
        for (auto& kv : rs.slot_of)
            if (!is_fe_val(kv.first)) used_slots = std::max(used_slots, (uint32_t)kv.second + 1);
    const uint32_t slot_bytes = (2 * (uint32_t)L + 4) * 512;
    const uint32_t unit_cap = std::min<uint32_t>(UNIT_MAX, (130u * 1024 - 8192 - used_slots * slot_bytes) / 1024);
    // one piece: gather records for its memory inputs, the prologue (if it restarts), the body
    auto emit_piece = [&](const Restart* rs, uint32_t pos_begin, uint32_t pos_end) {
        uint32_t piece_first = (uint32_t)out.size();
        StageMap sm;
        struct GEntry { uint32_t meta, ref; };
        std::vector<GEntry> gl;
        auto consider = [&](uint32_t pos, bool in_prologue) {
            const H2EOp& op = ops[alive[pos]];
            if ((op.flags & H2E_FLAG_HINTED) && kind_of(op) == K_MUL) {
                uint32_t key = op.imm | ((op.flags & H2E_FLAG_HINT_STRIDED) ? 0x80000000u : 0);
                if (sm.hint.count(key) || sm.units + HINT_UNITS > unit_cap) return;
                sm.hint[key] = sm.units;
                for (uint32_t hf = 0; hf < HINT_UNITS; hf++)
                    gl.push_back(GEntry{1u | (hf << 4) | ((op.flags & H2E_FLAG_HINT_STRIDED) ? 0x100u : 0), op.imm});
                sm.units += HINT_UNITS;
                return;
            }
            if (kind_of(op) == K_FULL || op.opcode == H2E_OP_PICK_INDEX) return;
            if (op.opcode == H2E_OP_SELECT_POINT && (op.flags & H2E_FLAG_PRESELECTED)) {
                if (sm.sel.count(op.refs[1]) || sm.units + 2 * HINT_UNITS > unit_cap) return;
                sm.sel[op.refs[1]] = sm.units;
                for (uint32_t which = 0; which < 2; which++)
                    for (uint32_t hf = 0; hf < HINT_UNITS; hf++)
                        gl.push_back(GEntry{2u | ((which * (H2E_W_WORDS_MAX / 2) + hf) << 4), op.refs[1]});
                sm.units += 2 * HINT_UNITS;
                return;
            }
            Opd o[3];
            int n = operands(op, o);
            for (int q = 0; q < n; q++) {
                bool external = dec[pos].val[q] < 0;
                (void)in_prologue;
                if (!external || !stageable(o[q].ref) || sm.cells.count(o[q].ref)) continue;
                uint32_t need = o[q].is_int ? INT_UNITS : FE_UNITS;
                bool ok = sm.units + need <= unit_cap;
                if (o[q].is_int)
                    for (int j = 0; j <= L; j++) ok = ok && stageable(op.refs[o[q].refpos + j]);
                if (!ok) continue;
                sm.cells[o[q].ref] = sm.units;
                if (o[q].is_int) {
                    for (int j = 0; j < L; j++) gl.push_back(GEntry{0u, op.refs[o[q].refpos + j]});   // low 16 bytes of a limb cell
                    gl.push_back(GEntry{0u, op.refs[o[q].refpos + L]});
                    gl.push_back(GEntry{0u | (1u << 4), op.refs[o[q].refpos + L]});
                } else {
                    gl.push_back(GEntry{0u, o[q].ref});
                    gl.push_back(GEntry{0u | (1u << 4), o[q].ref});
                }
                sm.units += need;
            }
        };
        if (rs)
            for (uint32_t q : rs->prologue) consider(q, true);
        for (uint32_t pos = pos_begin; pos < pos_end; pos++) consider(pos, false);
        // Staging (asynchronous gathers into LDS ahead of the serial chain) pays for segments with few lanes, whose
        // time is load latency; a segment with thousands of workgroups (the MSM windows: 254 strands x 38 pieces) is
        // bound by how many of them fit on a CU, and the staging area is half of its LDS (1.85 -> 0.85 ms).
        if (!stage_on || (uint64_t)sg->n_strands * (restarts.size() + 1) >= 2048) {
            sm = StageMap();
            gl.clear();
        }
        for (size_t e = 0; e < gl.size(); e += 3) {
            uint32_t n = (uint32_t)std::min<size_t>(3, gl.size() - e);
            H2EVRec g{{H2E_V_GATHER | (n << 8), (uint32_t)e, 0, 0, 0, 0, 0, 0}};
            for (uint32_t j = 0; j < n; j++) {
                g.w[2 + 2 * j] = gl[e + j].meta;
                g.w[3 + 2 * j] = gl[e + j].ref;
            }
            out.push_back(g);
        }
        if (!gl.empty()) out.push_back(H2EVRec{{H2E_V_GATHER_WAIT, 0, 0, 0, 0, 0, 0, 0}});
        max_units = std::max(max_units, sm.units);
        if (rs)
            for (uint32_t q : rs->prologue) emit(q, &rs->slot_of, sm);
        for (uint32_t pos = pos_begin; pos < pos_end; pos++) emit(pos, nullptr, sm);
        h_vpieces.push_back(vbase + piece_first);
        h_vpieces.push_back(vbase + (uint32_t)out.size());
        pad_chunk();
    };
    {
        uint32_t begin = 0;
        for (size_t ri = 0; ri <= restarts.size(); ri++) {
            uint32_t end = ri < restarts.size() ? restarts[ri].pos : (uint32_t)alive.size();
            emit_piece(ri == 0 ? nullptr : &restarts[ri - 1], begin, end);
            begin = end;
        }
    }
    seg_n_pieces[si] = (uint32_t)h_vpieces.size() / 2 - seg_piece_begin[si];
    seg_v_slots[si] = used_slots;
    seg_v_units[si] = std::max(1u, max_units);
    h_vtape.insert(h_vtape.end(), out.begin(), out.end());
    if (dbg_env("H2E_DUMP_TAPE"))
        fprintf(stderr, "segment %zu: replay %zu alive ops, %zu records, %u pieces, %u int slots, %u staging units\n", si, alive.size(),
                out.size(), seg_n_pieces[si], seg_v_slots[si], seg_v_units[si]);
}

