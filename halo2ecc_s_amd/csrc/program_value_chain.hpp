// h2e_program: liveness over sub-ranges, cross-segment escape analysis, field chains (field_chain.hpp) and hint stores of cut segments.
// Part of the C-ABI layer's one translation unit (included by h2e_capi.cpp behind program.hpp).
#pragma once

// Liveness over sub-ranges: an arithmetic op whose result cells are only read by ops of its own sub-range gets
// H2E_FLAG_LOCAL_RESULT, so the values-only replay keeps that result in LDS and does not store it (the full
// expansion of the sub-range recomputes and stores it anyway).  Row ownership: an op owns the rows from its
// first row up to the next op's first row.  Every reference that can reach a cut segment is considered: op
// refs of all segments, candidate tables (aux), strand parameters and the program's outputs.
void h2e_program::mark_local_results() {
    h2e::Recorder& r = *rec;
    // a sub-range's integer results must all fit the replay's LDS ring (VCache::R in engine.hip)
    const uint32_t ring = r.fp.limbs == 3 ? 20 : 16;
    auto puts = [](const H2EOp& op) -> uint32_t {
        switch (op.opcode) {
            case H2E_OP_SELECT_POINT: return 2;
            case H2E_OP_INT_ADD: case H2E_OP_INT_SUB: case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: case H2E_OP_INT_MUL:
            case H2E_OP_REDUCE: case H2E_OP_DIV_CORE: case H2E_OP_MASK_INT: case H2E_OP_BISEC_INT: return 1;
            default: return 0;
        }
    };
    struct CutSeg {
        const h2e::Segment* sg;
        H2EOp* ops;
        uint32_t n_ops;
        std::vector<uint32_t> sub_of;
        std::vector<uint8_t> escapes, sub_fits;
        uint32_t first[3], last[3];
    };
    std::vector<CutSeg> cs;
    for (auto& sg : r.segments) {
        uint32_t n_ops = sg.tape_end - sg.tape_begin;
        if (sg.n_cuts == 0 || n_ops == 0) continue;
        CutSeg c;
        c.sg = &sg;
        c.ops = r.tape.data() + sg.tape_begin;
        c.n_ops = n_ops;
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
        bool any = false;
        uint32_t prev = 0;
        for (uint32_t bnd : bounds) {
            uint32_t np = 0, nsingle = 0;
            for (uint32_t i = prev; i < bnd; i++) {
                np += puts(c.ops[i]);
                uint16_t oc = c.ops[i].opcode;
                if (oc == H2E_OP_IS_INT_ZERO || oc == H2E_OP_NOT || oc == H2E_OP_AND || oc == H2E_OP_OR || oc == H2E_OP_XNOR ||
                    oc == H2E_OP_PICK_INDEX)
                    nsingle++;
            }
            // too many results in a sub-range: all of its results stay stored
            c.sub_fits.push_back(np <= ring && nsingle <= 8);
            any = any || c.sub_fits.back();
            prev = bnd;
        }
        if (!any) continue;
        c.sub_of.assign(n_ops, 0);
        uint32_t sub = 0;
        for (uint32_t i = 0; i < n_ops; i++) {
            while (i >= bounds[sub]) sub++;
            c.sub_of[i] = sub;
        }
        c.escapes.assign(n_ops, 0);
        for (int reg = 0; reg < 3; reg++) {
            auto row_of = [&](uint32_t i) { return reg == 0 ? c.ops[i].base_row : reg == 1 ? c.ops[i].range_row : c.ops[i].select_row; };
            c.first[reg] = row_of(0);
            c.last[reg] = row_of(n_ops - 1) + 256;  // the last op writes < 256 rows
        }
        cs.push_back(std::move(c));
    }
    // producer of a row (strand-relative row for forks, absolute row for the main context) inside a cut segment
    auto producer = [&](const CutSeg& c, uint32_t region, uint32_t row) -> int {
        int lo = 0, hi = (int)c.n_ops - 1, ans = -1;
        while (lo <= hi) {
            int mid = (lo + hi) / 2;
            uint32_t first = region == 0 ? c.ops[mid].base_row : region == 1 ? c.ops[mid].range_row : c.ops[mid].select_row;
            if (first <= row) {
                ans = mid;
                lo = mid + 1;
            } else {
                hi = mid - 1;
            }
        }
        return ans;
    };
    // Field hints (Recorder::begin_field_hints): a segment whose mul-like ops carry hints of the field-domain predictor
    // keeps them only if the predictor can be compiled for it (every op in the value cone of the hinted ops is one it
    // knows, every operand a result of the segment itself); otherwise the flags go and the segment replays as before.
    auto field_compiler = [&](const CutSeg& c) {
        h2e::FieldCompiler fcmp;
        fcmp.ops = c.ops;
        fcmp.n_ops = c.n_ops;
        fcmp.L = r.fp.limbs;
        fcmp.pw_check_limbs = r.fp.pure_w_check_limbs;
        fcmp.w_words = r.fp.w_words;
        fcmp.rel = c.sg->is_fork ? 1 : 0;
        for (int reg = 0; reg < 3; reg++) {
            fcmp.first[reg] = c.first[reg];
            fcmp.last[reg] = c.last[reg];
        }
        const CutSeg* cp = &c;
        fcmp.producer = [cp, &producer](uint32_t region, uint32_t row) { return producer(*cp, region, row); };
        // H2E_FIELD_CHAIN=lanes: the one-lane-per-record kernel (A/B); default: a 16-lane row per record, 60 rows per pass
        const char* fm = dbg_env("H2E_FIELD_CHAIN");   // (debug-hook builds linked with -DH2E_AB_KERNELS engine units only: the product has no lane kernel)
        fcmp.digit_rows = !(fm && !strcmp(fm, "lanes"));
        return fcmp;
    };
    // A context cut into several segments with field hints (a pairing check: Miller loop | final exponentiation ...): an
    // integer a segment reads from an EARLIER one is found here - the producing segment and op - and enters the reader's
    // programs as an import (field_chain.hpp FieldCompiler::Import).
    struct FieldProducer {
        const CutSeg* c = nullptr;
        int op = -1;
    };
    std::vector<h2e::FieldCompiler> seg_fcmp;   // per cut segment: the producer lookups (no program state)
    for (auto& c : cs) seg_fcmp.push_back(field_compiler(c));
    auto find_field_producer = [&](const CutSeg& self, uint32_t ref) -> FieldProducer {
        FieldProducer none;
        if (ref == H2E_NO_REF || H2E_REF_REGION(ref) > 1 || H2E_REF_REL(ref)) return none;
        uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
        for (size_t k = 0; k < cs.size(); k++) {
            const CutSeg& c = cs[k];
            if (&c == &self) break;   // (cs is in segment order: earlier segments only)
            if (!c.sg->field_hints || c.sg->is_fork || row < c.first[region] || row >= c.last[region]) continue;
            int q = seg_fcmp[k].int_producer(ref);
            if (q >= 0) {
                FieldProducer fpd;
                fpd.c = &c;
                fpd.op = q;
                return fpd;
            }
        }
        return none;
    };
    typedef std::map<const CutSeg*, std::map<uint32_t, uint32_t>> ExportMap;   // producing segment -> op -> hint slot
    // the import callbacks of segment `self`: slots from `slots` (a value not in there yet - the analysis pass - reads as slot 0),
    // every imported value that needs a slot noted in `wanted`
    auto wire_imports = [&](h2e::FieldCompiler& fcmp, const CutSeg& self, const ExportMap* slots, ExportMap* wanted) {
        const CutSeg* sp = &self;
        fcmp.import_of = [sp, slots, &find_field_producer](uint32_t ref, h2e::FieldCompiler::Import& imp) -> bool {
            FieldProducer fpd = find_field_producer(*sp, ref);
            if (!fpd.c) return false;
            const H2EOp& po = fpd.c->ops[fpd.op];
            if (po.opcode == H2E_OP_CONST_INT) { imp.kind = 1; imp.imm = po.imm; return true; }
            if (po.opcode == H2E_OP_ASSIGN_W || po.opcode == H2E_OP_CONST_INT_INPUT) {
                if (po.flags & H2E_FLAG_INPUT_STRIDED) return false;
                imp.kind = 2;
                imp.imm = po.imm;
                return true;
            }
            imp.kind = 0;
            imp.imm = 0;
            if (slots) {
                auto it = slots->find(fpd.c);
                if (it != slots->end()) {
                    auto jt = it->second.find((uint32_t)fpd.op);
                    if (jt != it->second.end()) imp.imm = jt->second;
                }
            }
            return true;
        };
        fcmp.note_import = nullptr;
        if (wanted)
            fcmp.note_import = [sp, wanted, &find_field_producer](uint32_t ref) {
                FieldProducer fpd = find_field_producer(*sp, ref);
                if (!fpd.c) return;
                uint16_t oc = fpd.c->ops[fpd.op].opcode;
                if (oc == H2E_OP_CONST_INT || oc == H2E_OP_ASSIGN_W || oc == H2E_OP_CONST_INT_INPUT) return;
                (*wanted)[fpd.c][(uint32_t)fpd.op] = 0;
            };
    };
    // all field-hint segments of the program, last to first: can each one's chain and store be compiled, given what the
    // later ones want exported?  (One verdict for the group: a segment that falls back to a replay stores no hint slots
    // for the others to import.)
    auto analyse_field_segments = [&](ExportMap& wanted, bool with_store, std::string& why) -> bool {
        for (size_t k = cs.size(); k-- > 0;) {
            const CutSeg& c = cs[k];
            const h2e::Segment& sg = *c.sg;
            if (!sg.field_hints) continue;
            if (!(sg.n_strands == 1 && !sg.is_fork && sg.field_pair == r.fp.id) || getenv("H2E_NO_FIELD_CHAIN")) { why = "not a single-strand segment of the program's field"; return false; }
            h2e::FieldCompiler fcmp = field_compiler(c);
            wire_imports(fcmp, c, nullptr, &wanted);
            h2e::StoreCompiler sc;
            sc.ops = c.ops;
            sc.n_ops = c.n_ops;
            sc.L = r.fp.limbs;
            sc.fc = &r.fp.fc;
            sc.fcmp = &fcmp;
            sc.next_aux = 0;
            h2e::HintStore hs;
            if (with_store) {
                if (!sc.compile(hs)) { why = hs.why; return false; }
                fcmp.aux = &hs.aux_hint;
            } else if (!sc.feasible(why)) return false;
            fcmp.exports = &wanted[&c];
            h2e::FieldChain chain;
            if (!fcmp.compile(chain, true)) { why = chain.why; return false; }
        }
        return true;
    };
    {
        bool any = false, all_cut = true;
        for (auto& sg : r.segments) {
            if (!sg.field_hints) continue;
            any = true;
            bool cut = false;
            for (auto& c : cs) cut = cut || c.sg == &sg;
            all_cut = all_cut && cut;
        }
        if (any) {
            ExportMap wanted;
            std::string why = "segment has no cuts";
            bool ok = all_cut;
            try {
                ok = ok && analyse_field_segments(wanted, false, why);
            } catch (std::exception& e) {
                ok = false;
                why = e.what();
            }
            if (!ok)
                for (auto& sg : r.segments) {
                    if (!sg.field_hints) continue;
                    for (uint32_t i = sg.tape_begin; i < sg.tape_end; i++) {
                        H2EOp& op = r.tape[i];
                        if ((op.flags & H2E_FLAG_HINTED) && !(op.flags & H2E_FLAG_HINT_STRIDED) &&
                            (op.opcode == H2E_OP_INT_MUL || op.opcode == H2E_OP_REDUCE || op.opcode == H2E_OP_DIV_CORE))
                            op.flags &= ~(uint16_t)H2E_FLAG_HINTED;
                    }
                    sg.field_hints = false;
                    if (dbg_env("H2E_DUMP_TAPE")) fprintf(stderr, "segment %zu: no field chain (%s)\n", (size_t)(&sg - r.segments.data()), why.c_str());
                }
        }
    }
    if (cs.empty()) return;
    // an absolute reference from anywhere
    auto quote_abs = [&](uint32_t ref, const h2e::Segment* own = nullptr) {
        if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM || H2E_REF_REL(ref)) return;
        uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
        for (auto& c : cs) {
            const h2e::Segment& sg = *c.sg;
            if (&sg == own) continue;   // references inside the quoting segment itself are refined in step 1
            if (sg.is_fork) {
                uint32_t b0 = region == 0 ? sg.base0 : region == 1 ? sg.range0 : sg.select0;
                uint32_t d = region == 0 ? sg.dbase : region == 1 ? sg.drange : sg.dselect;
                if (d == 0 || row < b0 || row >= b0 + (uint64_t)d * sg.n_strands) continue;
                int pidx = producer(c, region, (row - b0) % d);
                if (pidx >= 0) c.escapes[pidx] = 1;
            } else {
                if (row < c.first[region] || row >= c.last[region]) continue;
                int pidx = producer(c, region, row);
                if (pidx >= 0) c.escapes[pidx] = 1;  // refined below for same-segment consumers
            }
        }
    };
    // 1. consumers inside the same segment (same addressing mode): only a different sub-range makes it escape
    for (auto& c : cs) {
        uint32_t rel = c.sg->is_fork ? 1 : 0;
        for (uint32_t i = 0; i < c.n_ops; i++)
            for (int k = 0; k < H2E_OP_MAX_REFS; k++) {
                uint32_t ref = c.ops[i].refs[k];
                if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM) continue;
                if (H2E_REF_REL(ref) != rel) continue;
                uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
                if (!rel && (row < c.first[region] || row >= c.last[region])) continue;
                int pidx = producer(c, region, row);
                if (pidx >= 0 && c.sub_of[pidx] != c.sub_of[i]) c.escapes[pidx] = 1;
            }
    }
    // 2. every absolute reference from other places
    for (auto& sg : r.segments) {
        bool is_cut_main = false;
        for (auto& c : cs)
            if (c.sg == &sg && !sg.is_fork) is_cut_main = true;
        // a cut main-context segment: its references into itself were handled in step 1, those into other cut
        // segments (the MSM tail reads the windows' sums) count like anybody else's
        for (uint32_t i = sg.tape_begin; i < sg.tape_end; i++)
            for (int k = 0; k < H2E_OP_MAX_REFS; k++) quote_abs(r.tape[i].refs[k], is_cut_main ? &sg : nullptr);
    }
    for (uint32_t ref : r.aux) quote_abs(ref);
    for (uint32_t ref : r.params) quote_abs(ref);
    for (uint32_t ref : r.outputs) quote_abs(ref);
    for (uint32_t ref : r.pre_args) quote_abs(ref);  // (small integers in there never alias region/row of a cut segment)
    // 3. flag
    auto is_arithmetic = [](uint16_t oc) {
        return oc == H2E_OP_INT_ADD || oc == H2E_OP_INT_SUB || oc == H2E_OP_INT_NEG || oc == H2E_OP_INT_MUL_SMALL ||
               oc == H2E_OP_INT_MUL || oc == H2E_OP_REDUCE || oc == H2E_OP_DIV_CORE || oc == H2E_OP_MASK_INT ||
               oc == H2E_OP_IS_INT_ZERO || oc == H2E_OP_NOT;
    };
    for (auto& c : cs)
        for (uint32_t i = 0; i < c.n_ops; i++)
            if (is_arithmetic(c.ops[i].opcode) && !c.escapes[i] && c.sub_fits[c.sub_of[i]]) c.ops[i].flags |= H2E_FLAG_LOCAL_RESULT;
    // 4. dead ops of the values-only replay: a local result that no op the replay has to execute reads.  What an
    // op reads *in values mode* (exec_op_values): a hinted INT_MUL / REDUCE / DIV_CORE reads nothing.
    const int L = r.fp.limbs;
    for (auto& c : cs) {
        uint32_t rel = c.sg->is_fork ? 1 : 0;
        std::vector<uint8_t> used(c.n_ops, 0);
        for (uint32_t i = c.n_ops; i-- > 0;) {
            H2EOp& op = c.ops[i];
            bool needed = !(op.flags & H2E_FLAG_LOCAL_RESULT) || used[i];
            if (op.opcode == H2E_OP_PICK_INDEX && (op.flags & H2E_FLAG_PRESELECTED)) needed = false;   // the select pre-kernel did it
            if (!needed) {
                op.flags |= H2E_FLAG_VALUES_SKIP;
                // a value hint nobody consumes needs neither checking nor producing (the division's quotient is
                // consumed by the expansion itself and keeps its hint)
                if ((op.flags & H2E_FLAG_HINTED) && (op.opcode == H2E_OP_INT_MUL || op.opcode == H2E_OP_REDUCE))
                    op.flags &= ~(uint16_t)(H2E_FLAG_HINTED | H2E_FLAG_HINT_STRIDED);
                continue;
            }
            uint32_t reads[H2E_OP_MAX_REFS];
            int nr = 0;
            bool hinted = (op.flags & H2E_FLAG_HINTED) != 0;
            switch (op.opcode) {
                case H2E_OP_INT_MUL:
                    if (!hinted) { reads[nr++] = op.refs[0]; reads[nr++] = op.refs[L + 1]; }
                    break;
                case H2E_OP_REDUCE:
                    if (!hinted) reads[nr++] = op.refs[0];
                    break;
                case H2E_OP_DIV_CORE:
                    if (!hinted) { reads[nr++] = op.refs[0]; reads[nr++] = op.refs[L + 1]; }
                    break;
                case H2E_OP_INT_ADD: case H2E_OP_INT_SUB:
                    reads[nr++] = op.refs[0]; reads[nr++] = op.refs[L + 1];
                    break;
                case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: case H2E_OP_IS_INT_ZERO:
                    reads[nr++] = op.refs[0];
                    break;
                case H2E_OP_MASK_INT:
                    reads[nr++] = op.refs[0]; reads[nr++] = op.refs[L + 1];
                    break;
                case H2E_OP_ASSERT_CONST: case H2E_OP_CACHE_INT: case H2E_OP_SUM_LIMBS:
                    break;   // nothing in values mode
                case H2E_OP_SELECT_POINT:
                    if (!(op.flags & H2E_FLAG_PRESELECTED)) reads[nr++] = op.refs[0];
                    break;
                default:     // everything else: every reference
                    for (int k = 0; k < H2E_OP_MAX_REFS; k++) reads[nr++] = op.refs[k];
                    break;
            }
            for (int k = 0; k < nr; k++) {
                uint32_t ref = reads[k];
                if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM || H2E_REF_REL(ref) != rel) continue;
                uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
                if (!rel && (row < c.first[region] || row >= c.last[region])) continue;
                int pidx = producer(c, region, row);
                if (pidx >= 0 && (uint32_t)pidx < i) used[pidx] = 1;
            }
        }
    }
    // 5. compile the values-only replay of every cut segment
    seg_piece_begin.assign(r.segments.size(), 0);
    seg_n_pieces.assign(r.segments.size(), 0);
    seg_v_slots.assign(r.segments.size(), 1);
    seg_v_units.assign(r.segments.size(), 1);
    seg_l_begin.assign(r.segments.size(), 0);
    seg_l_steps.assign(r.segments.size(), 0);
    seg_l_slots.assign(r.segments.size(), 0);
    seg_l_pair.assign(r.segments.size(), 0);
    seg_lr_begin.assign(r.segments.size(), 0);
    seg_l_recs.assign(r.segments.size(), 0);
    seg_s_begin.assign(r.segments.size(), 0);
    seg_so_begin.assign(r.segments.size(), 0);
    seg_sk_begin.assign(r.segments.size(), 0);
    seg_n_sops.assign(r.segments.size(), 0);
    // A segment whose mul-like results all have hints from the MSM predictors (the windows' strands, the tail) needs no
    // chain to put its escaping values in place either: every one of them is a hint or a limb-wise combination of hints,
    // pre-selected candidates and integers that were stored before the segment started (field_chain.hpp "hint store",
    // records with extension leaves).  Whatever does not fit that description keeps its compiled replay.
    seg_sx_begin.assign(r.segments.size(), 0);
    auto compile_plain_store = [&](const CutSeg& c) -> bool {
        const h2e::Segment& sg = *c.sg;
        if (getenv("H2E_NO_PLAIN_STORE") || sg.field_pair != r.fp.id) return false;
        const int L = r.fp.limbs;
        const uint32_t rel = sg.is_fork ? 1 : 0;
        h2e::FieldCompiler fcmp = field_compiler(c);
        h2e::HintStore hs;
        h2e::StoreCompiler sc;
        sc.ops = c.ops;
        sc.n_ops = c.n_ops;
        sc.L = L;
        sc.fc = &r.fp.fc;
        sc.fcmp = &fcmp;
        sc.next_aux = 0;
        typedef h2e::StoreCompiler::Lin Lin;
        std::vector<uint32_t> ext;
        std::map<std::array<uint32_t, H2E_SX_WORDS>, uint32_t> ext_index;
        bool overflow = false;
        auto ext_leaf = [&](const std::array<uint32_t, H2E_SX_WORDS>& e) -> uint32_t {
            auto it = ext_index.find(e);
            uint32_t idx;
            if (it != ext_index.end()) idx = it->second;
            else {
                idx = (uint32_t)(ext.size() / H2E_SX_WORDS);
                ext.insert(ext.end(), e.begin(), e.end());
                ext_index[e] = idx;
            }
            if (idx >= (1u << 22)) overflow = true;
            return (3u << 30) | (idx & 0x3fffffu);
        };
        auto small_leaf = [&](uint32_t kind, uint32_t index) -> uint32_t {
            if (index >= (1u << 22)) overflow = true;
            return (kind << 30) | (index & 0x3fffffu);
        };
        std::vector<Lin> lin(c.n_ops);
        std::vector<uint8_t> have(c.n_ops, 0);   // 1 done, 2 failed
        auto add_leaf = [](Lin& t, uint32_t leaf, int scale) {
            int& v = t.leaf[leaf];
            v += scale;
            if (v == 0) t.leaf.erase(leaf);
        };
        auto add_ceil = [](Lin& t, uint32_t times) {
            int& v = t.ceil[times];
            v += 1;
            if (v == 0) t.ceil.erase(times);
        };
        std::function<bool(uint32_t)> flatten = [&](uint32_t pi) -> bool {
            if (have[pi]) return have[pi] == 1;
            have[pi] = 2;
            const H2EOp& op = c.ops[pi];
            Lin rr;
            auto opd = [&](int refpos, int scale) -> bool {   // rr += scale * (the integer whose first limb cell is refs[refpos])
                uint32_t ref = op.refs[refpos];
                if (ref == H2E_NO_REF) return false;
                uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
                bool internal = region != H2E_REGION_PARAM && H2E_REF_REL(ref) == rel && (rel || (row >= c.first[region] && row < c.last[region]));
                if (internal && region <= 1) {
                    int q = fcmp.int_producer(ref);
                    if (q < 0 || (uint32_t)q >= pi || !flatten((uint32_t)q)) return false;
                    h2e::StoreCompiler::add_scaled(rr, lin[q], scale);
                    return true;
                }
                if (internal) {   // select rows: a coordinate of the point the select pre-kernel picked
                    int q = producer(c, 2, row);
                    if (q < 0 || (uint32_t)q >= pi) return false;
                    const H2EOp& so = c.ops[q];
                    if (so.opcode != H2E_OP_SELECT_POINT || !(so.flags & H2E_FLAG_PRESELECTED) || ((so.flags >> 8) & 0xffu) != 0 || H2E_REF_COL(ref) != 0) return false;
                    uint32_t off = row - so.select_row;
                    if (off != 0 && off != (uint32_t)L + 1) return false;
                    std::array<uint32_t, H2E_SX_WORDS> e{};
                    e[0] = H2E_SX_SEL;
                    e[1] = so.refs[1];
                    e[2] = off ? 1u : 0u;
                    add_leaf(rr, ext_leaf(e), scale);
                    return true;
                }
                // an integer of another segment (or reached through the strand's parameters): its cells were stored before this
                // segment's value chain started - the replay read them from there as well
                std::array<uint32_t, H2E_SX_WORDS> e{};
                e[0] = H2E_SX_CELLS;
                for (int i = 0; i <= L; i++) e[1 + i] = op.refs[refpos + i];
                add_leaf(rr, ext_leaf(e), scale);
                return true;
            };
            bool ok = true;
            switch (op.opcode) {
                case H2E_OP_INT_MUL: case H2E_OP_REDUCE: case H2E_OP_DIV_CORE:
                    if (!(op.flags & H2E_FLAG_HINTED)) { ok = false; break; }
                    if (op.flags & H2E_FLAG_HINT_STRIDED) {
                        std::array<uint32_t, H2E_SX_WORDS> e{};
                        e[0] = H2E_SX_HINT;
                        e[1] = op.imm;
                        add_leaf(rr, ext_leaf(e), 1);
                    } else add_leaf(rr, small_leaf(0, op.imm), 1);
                    break;
                case H2E_OP_CONST_INT: add_leaf(rr, small_leaf(1, op.imm), 1); break;
                case H2E_OP_ASSIGN_W: case H2E_OP_CONST_INT_INPUT:
                    if (op.flags & H2E_FLAG_INPUT_STRIDED) {
                        std::array<uint32_t, H2E_SX_WORDS> e{};
                        e[0] = H2E_SX_INPUT;
                        e[1] = op.imm;
                        add_leaf(rr, ext_leaf(e), 1);
                    } else add_leaf(rr, small_leaf(2, op.imm), 1);
                    break;
                case H2E_OP_INT_ADD: ok = opd(0, 1) && opd(L + 1, 1); break;
                case H2E_OP_INT_SUB:   // a - b + C_(b.times)   (integer_chip.rs:408-437)
                    ok = opd(0, 1) && opd(L + 1, -1);
                    add_ceil(rr, op.imm);
                    break;
                case H2E_OP_INT_NEG:   // C_(a.times) - a       (:439-464)
                    ok = opd(0, -1);
                    add_ceil(rr, op.imm);
                    break;
                case H2E_OP_INT_MUL_SMALL: ok = opd(0, (int)op.imm); break;
                default: ok = false; break;
            }
            if (!ok) return false;
            lin[pi] = std::move(rr);
            have[pi] = 1;
            return true;
        };
        {
            Lin zero;
            sc.k_of(hs, zero);   // entry 0 of the K table
        }
        auto emit = [&](uint32_t kind, uint32_t k_idx, uint32_t w1, uint32_t w2, const std::vector<uint32_t>& terms) -> bool {
            if (terms.size() > 255 || k_idx > 0xffff) return false;
            hs.offsets.push_back((uint32_t)hs.words.size());
            hs.words.push_back(kind | ((uint32_t)terms.size() << 8) | (k_idx << 16));
            hs.words.push_back(w1);
            hs.words.push_back(w2);
            hs.words.insert(hs.words.end(), terms.begin(), terms.end());
            return true;
        };
        auto term = [](uint32_t leaf_word, int coef) -> uint32_t { return (leaf_word & 0xc0000000u) | ((uint32_t)(coef + 128) << 22) | (leaf_word & 0x3fffffu); };
        for (uint32_t i = 0; i < c.n_ops; i++) {
            const H2EOp& op = c.ops[i];
            if (op.flags & H2E_FLAG_VALUES_SKIP) continue;
            const bool stored = c.escapes[i] || !c.sub_fits[c.sub_of[i]];
            switch (op.opcode) {
                case H2E_OP_NOP: case H2E_OP_ASSERT_CONST: case H2E_OP_SUM_LIMBS: case H2E_OP_CACHE_INT: break;   // nothing in values mode
                case H2E_OP_PICK_INDEX:
                    if (!(op.flags & H2E_FLAG_PRESELECTED) && stored) return false;
                    break;
                case H2E_OP_SELECT_POINT:
                    if (!(op.flags & H2E_FLAG_PRESELECTED) || stored) return false;
                    break;
                case H2E_OP_ASSIGN_W: case H2E_OP_ASSIGN: case H2E_OP_ASSIGN_BIT: case H2E_OP_CONST: case H2E_OP_CONST_INT_INPUT:
                    if (!emit(H2E_S_FULL, 0, i, 0, {})) return false;   // ops without operands: run as they are
                    break;
                case H2E_OP_CONST_INT:
                    if (stored && !emit(H2E_S_CONST, 0, op.base_row, 0, {term(small_leaf(1, op.imm), 1)})) return false;
                    break;
                case H2E_OP_INT_MUL: case H2E_OP_REDUCE: case H2E_OP_DIV_CORE: {
                    if (!stored) break;
                    if (!flatten(i)) return false;
                    if (!emit(H2E_S_W, 0, op.base_row, op.range_row, {term(lin[i].leaf.begin()->first, 1)})) return false;
                } break;
                case H2E_OP_INT_ADD: case H2E_OP_INT_SUB: case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: {
                    if (!stored) break;
                    if (!flatten(i)) return false;
                    std::vector<uint32_t> terms;
                    long weight = 1;
                    for (auto& kv : lin[i].leaf) {
                        if (kv.second < -127 || kv.second > 127) return false;
                        terms.push_back(term(kv.first, kv.second));
                        weight += std::abs(kv.second);
                    }
                    if (weight >= 4096) return false;
                    if (!emit(H2E_S_LIN, sc.k_of(hs, lin[i]), op.base_row, 0, terms)) return false;
                } break;
                default:   // conditions, selections, decompositions ...: only if nothing outside their sub-range reads them
                    if (stored) return false;
                    break;
            }
        }
        if (overflow || hs.offsets.empty()) return false;
        size_t si = (size_t)(c.sg - r.segments.data());
        seg_s_begin[si] = (uint32_t)h_swords.size();
        seg_so_begin[si] = (uint32_t)h_soffsets.size();
        seg_sk_begin[si] = (uint32_t)h_sktab.size();
        seg_sx_begin[si] = (uint32_t)h_sext.size();
        seg_n_sops[si] = (uint32_t)hs.offsets.size();
        h_swords.insert(h_swords.end(), hs.words.begin(), hs.words.end());
        h_soffsets.insert(h_soffsets.end(), hs.offsets.begin(), hs.offsets.end());
        h_sktab.insert(h_sktab.end(), hs.ktab.begin(), hs.ktab.end());
        h_sext.insert(h_sext.end(), ext.begin(), ext.end());
        if (dbg_env("H2E_DUMP_TAPE"))
            fprintf(stderr, "segment %zu: plain hint store in place of the replay: %zu store ops per strand, %zu words, %zu extension leaves\n", si,
                    hs.offsets.size(), hs.words.size(), ext.size() / H2E_SX_WORDS);
        return true;
    };
    for (auto& c : cs)
        if (!c.sg->field_hints && !compile_plain_store(c))
            compile_replay(c.sg, c.ops, c.n_ops, c.first, c.last, [&](uint32_t region, uint32_t row) { return producer(c, region, row); });
    // 6. segments with field hints: the hint store in place of a replay, and the field-domain predictor whose program
    // goes into the pre-kernel args.  First (last segment to first) what each segment must leave in hint slots for the
    // later ones, then the programs themselves, first to last, every segment's compile-time slots (exports, conditions,
    // sink terms) in one block of their own.
    ExportMap wanted;
    {
        bool any = false;
        for (auto& c : cs) any = any || c.sg->field_hints;
        std::string why;
        if (any && !analyse_field_segments(wanted, true, why)) throw std::runtime_error("field chain: " + why);
    }
    ExportMap export_slots;
    for (auto& c : cs) {
        if (!c.sg->field_hints) continue;
        size_t si = (size_t)(c.sg - r.segments.data());
        const uint32_t hint_split = r.n_hint_slots;
        {
            std::map<uint32_t, uint32_t>& mine = export_slots[&c];
            for (auto& kv : wanted[&c]) {
                const H2EOp& op = c.ops[kv.first];
                bool own = (op.flags & H2E_FLAG_HINTED) && !(op.flags & H2E_FLAG_HINT_STRIDED) &&
                           (op.opcode == H2E_OP_INT_MUL || op.opcode == H2E_OP_REDUCE || op.opcode == H2E_OP_DIV_CORE);
                mine[kv.first] = own ? op.imm : r.n_hint_slots++;
            }
        }
        h2e::FieldCompiler fcmp = field_compiler(c);
        wire_imports(fcmp, c, &export_slots, nullptr);
        h2e::HintStore hs;
        {
            h2e::StoreCompiler sc;
            sc.ops = c.ops;
            sc.n_ops = c.n_ops;
            sc.L = r.fp.limbs;
            sc.fc = &r.fp.fc;
            sc.fcmp = &fcmp;
            sc.next_aux = r.n_hint_slots;
            if (!sc.compile(hs)) throw std::runtime_error(hs.why);
            r.n_hint_slots = sc.next_aux;
            seg_s_begin[si] = (uint32_t)h_swords.size();
            seg_so_begin[si] = (uint32_t)h_soffsets.size();
            seg_sk_begin[si] = (uint32_t)h_sktab.size();
            seg_sx_begin[si] = (uint32_t)h_sext.size();
            seg_n_sops[si] = (uint32_t)hs.offsets.size();
            h_swords.insert(h_swords.end(), hs.words.begin(), hs.words.end());
            h_soffsets.insert(h_soffsets.end(), hs.offsets.begin(), hs.offsets.end());
            h_sktab.insert(h_sktab.end(), hs.ktab.begin(), hs.ktab.end());
            h_sext.insert(h_sext.end(), hs.ext.begin(), hs.ext.end());
            if (dbg_env("H2E_DUMP_TAPE"))
                fprintf(stderr, "segment %zu: hint store: %zu store ops, %zu words, %zu K constants, at most %u terms, %zu aux hint slots\n", si,
                        hs.offsets.size(), hs.words.size(), hs.ktab.size() / (2 * (size_t)r.fp.limbs + 4), hs.n_terms_max, hs.aux_hint.size());
        }
        fcmp.aux = &hs.aux_hint;
        fcmp.exports = &export_slots[&c];
        fcmp.hint_split = hint_split;
        uint32_t next_hint = r.n_hint_slots;
        fcmp.next_hint = &next_hint;
        h2e::FieldChain chain;
        if (!fcmp.compile(chain, false)) throw std::runtime_error("field chain: " + chain.why);
        r.n_hint_slots = next_hint;
        h2e::PreKernel pk;
        std::memset(&pk.k, 0, sizeof(pk.k));
        pk.k.kind = H2E_PRE_FIELD_CHAIN;
        pk.k.n_lanes = 1;
        pk.k.hint_base = chain.hint_hi > chain.hint_lo ? chain.hint_lo : 0;
        pk.k.hints_per_lane = chain.hint_hi > chain.hint_lo ? chain.hint_hi - chain.hint_lo : 0;
        pk.k.hint2_base = chain.hint2_hi > chain.hint2_lo ? chain.hint2_lo : 0;
        pk.k.hints2_per_lane = chain.hint2_hi > chain.hint2_lo ? chain.hint2_hi - chain.hint2_lo : 0;
        pk.k.n_params = (uint32_t)r.fp.w_words;   // words per input slot
        while (r.pre_args.size() % 16) r.pre_args.push_back(0);   // records are read 16 bytes at a time from 64-byte aligned chunks
        pk.k.f_recs = (uint32_t)r.pre_args.size();
        pk.k.f_n_recs = (uint32_t)(chain.recs.size() / chain.rec_words);
        r.pre_args.insert(r.pre_args.end(), chain.recs.begin(), chain.recs.end());
        pk.k.f_rounds = (uint32_t)r.pre_args.size();
        pk.k.f_n_rounds = (uint32_t)(chain.rounds.size() / 2);
        r.pre_args.insert(r.pre_args.end(), chain.rounds.begin(), chain.rounds.end());
        pk.k.f_slots = chain.n_slots;
        pk.k.f_n_load_rounds = chain.n_load_rounds;
        pk.k.f_mode = fcmp.digit_rows ? 1 : 0;
        pk.k.f_sinks = (uint32_t)r.pre_args.size();
        r.pre_args.insert(r.pre_args.end(), chain.sink_offsets.begin(), chain.sink_offsets.end());
        pk.k.f_sink_words = (uint32_t)r.pre_args.size();
        r.pre_args.insert(r.pre_args.end(), chain.sink_words.begin(), chain.sink_words.end());
        pk.k.f_n_sinks = (uint32_t)chain.sink_offsets.size();
        pk.before_segment = (uint32_t)(c.sg - r.segments.data());
        pk.early_after_segment = -1;
        r.pre_kernels.push_back(pk);
        if (dbg_env("H2E_DUMP_TAPE") || getenv("H2E_FIELD_STATS"))
            fprintf(stderr, "segment %u: field chain: %u nodes (%u products, %u linear combinations), %u rounds, %u value slots, hint slots [%u, %u) + [%u, %u), %zu exports\n",
                    pk.before_segment, chain.n_nodes, chain.n_mul, chain.n_lin, pk.k.f_n_rounds, chain.n_slots, chain.hint_lo, chain.hint_hi,
                    chain.hint2_lo, chain.hint2_hi, export_slots[&c].size());
    }
}

