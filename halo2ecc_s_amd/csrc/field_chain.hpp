// Field-domain predictor chain ("field hints", Recorder::hint_mode 3): host compiler.
//
// A pairing check is ~175 k integer-chip ops whose dependency graph is 8.5 k levels deep - but only ~650 of those levels
// are products.  Everything between two products is linear (int_add / int_sub / int_neg / int_mul_small_constant,
// src/circuit/integer_chip.rs:384-464, :618-658) or a `reduce` (:283-373), and as *values mod w* a reduce is the identity
// and a chain of additions is one linear combination: the `find_w_modulus_of_ceil_times` constants int_sub / int_neg add
// (range_info.rs:334-359) are multiples of w.  So the canonical value of every mul-like result (what the values-only
// replay needs as a hint, tape.h H2E_FLAG_HINTED) is computed by a much shorter program over plain residues mod w in
// Montgomery form: one round of linear combinations, one round of Montgomery products, and so on - ~1.3 k rounds instead
// of 16 k.  This file turns a recorded segment into that program: records of 8 words, rounds of up to 64 independent
// records (one wave, engine.hip h2e_field_chain), values in LDS slots.
//
// Node kinds (= record opcodes):
//   LIN      dst = sum of up to F_MAX_TERMS terms coef * slot (coef a small signed integer)
//   MUL      dst = a * b                       DIV   dst = a / b (0 if b = 0: integer_chip.rs:524-527)
//   ISZERO   cond = (a == 0)                   NOT / AND / OR / XNOR on conditions (0 / 1)
//   SELECT   dst = cond ? a : b (b = none: 0)  (int_div's mask :511-520, bisec_int :660-681)
//   INPUT_W / CONST_W    a canonical W value from the instance inputs / the constant pool -> Montgomery form
//   INPUT_FE / CONST_FE  a condition from the inputs (assign_bit) / the pool (assign_constant)
// A record with a hint slot also stores its value (Montgomery form) into the hint workspace; h2e_field_finalize turns
// the slots into canonical values afterwards.
#pragma once
#include <array>
#include <algorithm>
#include <functional>
#include <map>
#include <stdexcept>
#include <vector>
#include <cstdlib>
#include <cstdio>
#include <string>
#include "tape.h"
#include "hbig.hpp"

namespace h2e {

struct FieldChain {
    std::vector<uint32_t> recs;     // rec_words (8 or 16) words per record, no round straddles an H2E_WCHUNK-record chunk
    uint32_t rec_words = 8;
    std::vector<uint32_t> rounds;   // per round: first record, count | kind << 8
    uint32_t n_slots = 0, n_load_rounds = 0;
    uint32_t n_nodes = 0, n_mul = 0, n_lin = 0;
    uint32_t hint_lo = 0xffffffffu, hint_hi = 0;   // hint slots written: [hint_lo, hint_hi)
    uint32_t hint2_lo = 0xffffffffu, hint2_hi = 0; // ... and those at or above FieldCompiler::hint_split (slots taken at compile time: with several
                                                   // segments another segment's recorded slots lie between the two ranges, and a finalize must not touch them)
    void note_hint(uint32_t h, uint32_t split) {
        if (h >= split) {
            hint2_lo = std::min(hint2_lo, h);
            hint2_hi = std::max(hint2_hi, h + 1);
        } else {
            hint_lo = std::min(hint_lo, h);
            hint_hi = std::max(hint_hi, h + 1);
        }
    }
    // hint-only linear combinations computed outside the chain (h2e_field_sinks): per sink its first word in sink_words;
    // a sink = hint slot, terms, then per term (coef & 0x1ff) << 23 | kind << 21 | index  (kind 0 hint slot, 1 input slot, 2 pool word)
    std::vector<uint32_t> sink_offsets, sink_words;
    std::string why;                // why a segment is not eligible
};

struct FieldCompiler {
    const H2EOp* ops;
    uint32_t n_ops;
    int L;
    int pw_check_limbs;
    int w_words = 4;                                   // 64-bit words of a W value (a value slot)
    std::function<int(uint32_t, uint32_t)> producer;   // (region, row) -> op index of this segment or -1
    uint32_t rel;                                      // 1: the segment's own cells are strand-relative refs
    uint32_t first[3], last[3];
    // further results the hint store wants in hint slots (conditions as raw 0 / 1, masked integers): op index -> slot
    const std::map<uint32_t, uint32_t>* aux = nullptr;
    uint32_t* next_hint = nullptr;                     // further hint slots may be taken from here (values the sinks kernel reads)
    bool digit_rows = true;                            // for h2e_field_chain_digits: rounds of at most 60 records sorted by opcode, 16-word records
    // A context cut into several segments with field hints (a pairing check: Miller loop | final exponentiation, so that one
    // segment's expansion runs under the next one's chain): an integer operand produced by an EARLIER segment enters this
    // segment's program as a load - of the hint slot the producer's chain left its value in (canonical once that segment's
    // finalize has run; `exports` tells the producer which of its results must get one), or of the constant / input the
    // producer itself loaded.  import_of resolves such an operand (false: nobody knows it), note_import reports every one
    // the value cone reaches (the analysis pass that collects the earlier segments' exports).
    struct Import {
        uint32_t kind;   // 0: hint slot, 1: pool constant (word offset), 2: input slot
        uint32_t imm;
    };
    std::function<bool(uint32_t ref, Import&)> import_of;
    std::function<void(uint32_t ref)> note_import;
    const std::map<uint32_t, uint32_t>* exports = nullptr;   // op index -> hint slot its value must be left in
    uint32_t hint_split = 0xffffffffu;                       // hint slots from here on were taken at compile time (FieldChain::hint2_lo / hi)

    struct Node {
        uint8_t opc = 0;
        int a = -1, b = -1, c = -1;                       // operand nodes
        std::vector<std::pair<int, int>> terms;           // LIN: (node, coef)
        uint32_t imm = 0;                                 // input slot / pool offset
        bool hint_src = false;                            // F_INPUT_W: imm is a hint slot of an earlier segment (record flag 0x100)
        uint32_t hint = 0xffffffffu;
    };
    std::vector<Node> nodes;
    struct Expr {
        std::vector<std::pair<int, int>> t;               // (base node, coef), sorted by node
    };

    static bool is_int_op(uint16_t oc) {
        switch (oc) {
            case H2E_OP_INT_MUL: case H2E_OP_REDUCE: case H2E_OP_DIV_CORE: case H2E_OP_INT_ADD: case H2E_OP_INT_SUB: case H2E_OP_INT_NEG:
            case H2E_OP_INT_MUL_SMALL: case H2E_OP_MASK_INT: case H2E_OP_BISEC_INT: case H2E_OP_CONST_INT: case H2E_OP_CONST_INT_INPUT:
            case H2E_OP_ASSIGN_W: return true;
            default: return false;
        }
    }
    static bool is_cond_op(uint16_t oc) {
        switch (oc) {
            case H2E_OP_IS_INT_ZERO: case H2E_OP_NOT: case H2E_OP_AND: case H2E_OP_OR: case H2E_OP_XNOR: case H2E_OP_ASSIGN_BIT:
            case H2E_OP_CONST: case H2E_OP_ASSIGN: return true;
            default: return false;
        }
    }
    uint32_t fe_row(const H2EOp& op) const {
        if (op.opcode == H2E_OP_IS_INT_ZERO) return op.base_row + 6 + 4 * (uint32_t)pw_check_limbs;
        return op.base_row;
    }
    // the op of this segment whose integer / condition result starts at `ref`, or -1
    int int_producer(uint32_t ref) const {
        if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM || H2E_REF_REL(ref) != rel) return -1;
        uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref), col = H2E_REF_COL(ref);
        if (region > 1) return -1;
        if (!rel && (row < first[region] || row >= last[region])) return -1;
        int p = producer(region, row);
        if (p < 0) return -1;
        const H2EOp& po = ops[p];
        switch (po.opcode) {
            case H2E_OP_INT_MUL: case H2E_OP_REDUCE: case H2E_OP_DIV_CORE: case H2E_OP_ASSIGN_W:
                return (region == 1 && col == 0 && row == po.range_row) ? p : -1;
            case H2E_OP_INT_ADD: case H2E_OP_INT_SUB: case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: case H2E_OP_MASK_INT: case H2E_OP_BISEC_INT:
                return (region == 0 && col == 4 && row == po.base_row) ? p : -1;
            case H2E_OP_CONST_INT: case H2E_OP_CONST_INT_INPUT:
                return (region == 0 && col == 0 && row == po.base_row) ? p : -1;
            default: return -1;
        }
    }
    int cond_producer(uint32_t ref) const {
        if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM || H2E_REF_REL(ref) != rel) return -1;
        uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref), col = H2E_REF_COL(ref);
        if (region != 0) return -1;
        if (!rel && (row < first[0] || row >= last[0])) return -1;
        int p = producer(region, row);
        if (p < 0) return -1;
        const H2EOp& po = ops[p];
        switch (po.opcode) {
            case H2E_OP_IS_INT_ZERO: case H2E_OP_NOT: case H2E_OP_AND: case H2E_OP_OR: case H2E_OP_XNOR:
                return (col == 4 && row == fe_row(po)) ? p : -1;
            case H2E_OP_ASSIGN_BIT: case H2E_OP_CONST: case H2E_OP_ASSIGN:
                return (col == 0 && row == po.base_row) ? p : -1;
            default: return -1;
        }
    }
    struct Opd { int refpos; bool is_int; };
    int operands(const H2EOp& op, Opd* o) const {
        int n = 0;
        switch (op.opcode) {
            case H2E_OP_INT_MUL: case H2E_OP_DIV_CORE: case H2E_OP_INT_ADD: case H2E_OP_INT_SUB:
                o[n++] = {0, true}; o[n++] = {L + 1, true}; break;
            case H2E_OP_REDUCE: case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: case H2E_OP_IS_INT_ZERO:
                o[n++] = {0, true}; break;
            case H2E_OP_MASK_INT:
                o[n++] = {0, true}; o[n++] = {L + 1, false}; break;
            case H2E_OP_BISEC_INT:
                o[n++] = {0, false}; o[n++] = {1, true}; o[n++] = {L + 2, true}; break;
            case H2E_OP_NOT:
                o[n++] = {0, false}; break;
            case H2E_OP_AND: case H2E_OP_OR: case H2E_OP_XNOR:
                o[n++] = {0, false}; o[n++] = {1, false}; break;
            default: break;
        }
        return n;
    }

    int F_MAX_TERMS = 6;   // terms of a LIN record: its words less two (compile() sets it)
    static constexpr int F_MAX_COEF = 255;
    enum { F_NOP = 0, F_LIN, F_MUL, F_DIV, F_ISZERO, F_NOT, F_AND, F_OR, F_XNOR, F_SELECT, F_INPUT_W, F_INPUT_FE, F_CONST_W, F_CONST_FE, F_RESERVED_14, F_CONT };

    int new_node(uint8_t opc) {
        nodes.emplace_back();
        nodes.back().opc = opc;
        return (int)nodes.size() - 1;
    }
    static int max_coef(const Expr& e) {
        int m = 0;
        for (auto& kv : e.t) m = std::max(m, std::abs(kv.second));
        return m;
    }
    static Expr merged(const Expr& a, int sa, const Expr& b, int sb) {
        std::map<int, long long> acc;
        for (auto& kv : a.t) acc[kv.first] += (long long)sa * kv.second;
        for (auto& kv : b.t) acc[kv.first] += (long long)sb * kv.second;
        Expr r;
        for (auto& kv : acc)
            if (kv.second != 0) r.t.push_back({kv.first, (int)std::max<long long>(-(1 << 30), std::min<long long>(1 << 30, kv.second))});
        return r;
    }
    bool fits(const Expr& e) const { return (int)e.t.size() <= F_MAX_TERMS && max_coef(e) <= F_MAX_COEF; }
    static Expr single(int node) {
        Expr e;
        e.t.push_back({node, 1});
        return e;
    }
    std::vector<Expr> expr;          // per op: its integer result as a linear combination of nodes
    // the node that holds the value of op p's result; from then on the op's expression is that node (every consumer
    // shares it, and later combinations start from one term)
    int mat(int p) {
        Expr& e = expr[p];
        if (e.t.size() == 1 && e.t[0].second == 1) return e.t[0].first;
        int n = new_node(F_LIN);
        nodes[n].terms = e.t;   // (an empty expression is the value 0)
        e = single(n);
        return n;
    }
    // sa * result(pa) + sb * result(pb) (pb < 0: no second operand), materialising operands when the combination would
    // not fit a LIN record
    Expr combine(int pa, int sa, int pb, int sb) {
        static const Expr none;
        auto eb = [&]() -> const Expr& { return pb >= 0 ? expr[pb] : none; };
        Expr r = merged(expr[pa], sa, eb(), sb);
        if (fits(r)) return r;
        bool a_first = pb < 0 || expr[pa].t.size() >= expr[pb].t.size();
        mat(a_first ? pa : pb);
        r = merged(expr[pa], sa, eb(), sb);
        if (fits(r)) return r;
        if (pb >= 0) mat(a_first ? pb : pa);
        r = merged(expr[pa], sa, eb(), sb);
        if (!fits(r)) throw std::runtime_error("field chain: a two-term combination does not fit a record");
        return r;
    }

    // The depth a combination of terms with these depths (sorted, deepest first) can be had at: a record of its own if they
    // fit, otherwise partial sums over the earliest terms until F_MAX_TERMS entries are left.
    // (KF: entries the final record may have - 2 F_MAX_TERMS for a LONG combination, whose second record sits behind the round's rows)
    uint32_t tree_depth(std::vector<uint32_t> d, size_t KF) const {
        const size_t K = (size_t)F_MAX_TERMS;
        if (d.empty()) return 1;
        while (d.size() > KF) {   // (d ascending here: the earliest terms at the front)
            size_t m = std::min(K, d.size() - (KF - 1));
            uint32_t pd = d[m - 1] + 1;
            d.erase(d.begin(), d.begin() + (long)m);
            d.insert(std::upper_bound(d.begin(), d.end(), pd), pd);
        }
        return d.back() + 1;
    }
    bool long_lins = true;   // combinations of up to 2 F_MAX_TERMS terms (a second record; H2E_FIELD_NO_LONG=1: off, A/B)
    void rebalance(bool sinks_enabled) {
        const size_t N0 = nodes.size();
        const size_t K = (size_t)F_MAX_TERMS;
        const size_t KL = long_lins ? 2 * K : K;
        auto best_depth = [&](const std::vector<uint32_t>& d) { return std::min(tree_depth(d, K), tree_depth(d, KL)); };
        std::vector<uint32_t> depth(N0, 0);
        depth.reserve(N0 * 2);
        std::map<std::vector<std::pair<int, int>>, int> partial_of;   // terms (sorted by node) -> the partial-sum node that holds them
        auto dep_depth = [&](const Node& nd) {
            uint32_t dd = 0;
            auto see = [&](int x) { if (x >= 0) dd = std::max(dd, depth[(size_t)x]); };
            see(nd.a);
            see(nd.b);
            see(nd.c);
            for (auto& t : nd.terms) see(t.first);
            return dd;
        };
        auto depths_of = [&](const std::map<int, long long>& E) {
            std::vector<uint32_t> d;
            d.reserve(E.size());
            for (auto& kv : E) d.push_back(depth[(size_t)kv.first]);
            std::sort(d.begin(), d.end());
            return d;
        };
        typedef std::vector<std::pair<uint32_t, std::pair<int, int>>> TermsByDepth;   // (depth, (node, coef)), ascending by depth
        // partial sums over the earliest terms until KF entries are left
        auto merge_earliest = [&](TermsByDepth& T, size_t KF) {
            while (T.size() > KF) {
                size_t m = std::min(K, T.size() - (KF - 1));
                std::vector<std::pair<int, int>> g;
                uint32_t gd = 0;
                for (size_t q = 0; q < m; q++) {
                    g.push_back(T[q].second);
                    gd = std::max(gd, T[q].first);
                }
                std::sort(g.begin(), g.end());
                int pn;
                auto it = partial_of.find(g);
                if (it != partial_of.end()) {
                    pn = it->second;
                } else {
                    pn = new_node(F_LIN);
                    nodes[(size_t)pn].terms = g;
                    depth.push_back(gd + 1);
                    partial_of[g] = pn;
                }
                T.erase(T.begin(), T.begin() + (long)m);
                std::pair<uint32_t, std::pair<int, int>> e{depth[(size_t)pn], {pn, 1}};
                T.insert(std::upper_bound(T.begin(), T.end(), e, [](const auto& x, const auto& y) { return x.first < y.first; }), e);
            }
        };
        for (size_t k = 0; k < N0; k++) {
            if (nodes[k].opc != F_LIN) {
                depth[k] = dep_depth(nodes[k]) + 1;
                continue;
            }
            std::map<int, long long> E;
            for (auto& t : nodes[k].terms) E[t.first] += t.second;
            // expand the whole deepest level at a time (two operands of a sum are usually equally deep: opening one of them alone
            // gains nothing), keep the best expression seen, give up after a few levels that did not help
            uint32_t best = best_depth(depths_of(E));
            {
                std::map<int, long long> cur = E;
                int stale = 0;
                for (int iter = 0; iter < 64 && stale < 3; iter++) {
                    uint32_t dm = 0;
                    for (auto& kv : cur) dm = std::max(dm, depth[(size_t)kv.first]);
                    bool open = !cur.empty();
                    for (auto& kv : cur)
                        if (depth[(size_t)kv.first] == dm && (nodes[(size_t)kv.first].opc != F_LIN || nodes[(size_t)kv.first].terms.empty())) open = false;
                    if (!open) break;   // a product (or an input) is what makes this level deep
                    std::map<int, long long> nxt;
                    bool ok = true;
                    for (auto& kv : cur) {
                        if (depth[(size_t)kv.first] != dm) {
                            nxt[kv.first] += kv.second;
                            continue;
                        }
                        for (auto& u : nodes[(size_t)kv.first].terms) nxt[u.first] += kv.second * u.second;
                    }
                    for (auto it = nxt.begin(); it != nxt.end();) {
                        if (std::llabs(it->second) > F_MAX_COEF) ok = false;
                        it = it->second == 0 ? nxt.erase(it) : std::next(it);
                    }
                    if (!ok || nxt.size() > 4 * K * K) break;
                    cur = std::move(nxt);
                    uint32_t d2 = best_depth(depths_of(cur));
                    if (d2 < best || (d2 == best && cur.size() <= K && cur.size() <= E.size())) {
                        E = cur;
                        best = d2;
                        stale = 0;
                    } else {
                        stale++;
                    }
                }
            }
            // the tree: partial sums over the earliest terms
            TermsByDepth T;
            for (auto& kv : E)
                if (kv.second != 0) T.push_back({depth[(size_t)kv.first], {kv.first, (int)kv.second}});
            std::stable_sort(T.begin(), T.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
            // a long final record only where it makes the value available a level earlier, and only while the columns' bias covers
            // its coefficients (sum |coef| <= 14 x 255, the bound of a plain record: H2EFieldConsts::lin_bias)
            size_t KF = K;
            {
                std::vector<uint32_t> dd = depths_of(E);
                if (KL > K && tree_depth(dd, KL) < tree_depth(dd, K)) {
                    long long abs_sum = 0;
                    size_t n_direct = std::min(T.size(), KL);   // (an upper bound: the latest terms stay in the final record, partial sums enter with coefficient 1)
                    for (size_t q = T.size() - n_direct; q < T.size(); q++) abs_sum += std::llabs((long long)T[q].second.second);
                    if (abs_sum + (long long)KL <= (long long)F_MAX_COEF * (long long)K) KF = KL;
                }
            }
            merge_earliest(T, KF);
            std::vector<std::pair<int, int>> r;
            for (auto& e : T) r.push_back(e.second);
            std::sort(r.begin(), r.end());
            nodes[k].terms = r;
            depth[k] = dep_depth(nodes[k]) + 1;
        }
        // (Tried in round 5 and taken out again: a second pass that turned long combinations OFF a critical path - latest levels from the
        // hinted nodes backwards - back into plain records; 18-55 % of them, rounds holding one 333 -> 273 and 339 -> 245 of a bn256 check's
        // - and no measurable change of the chain, gpurun_out/r5_18: a round is not as long as its fattest row, it is as long as its two passes.)
        // hinted combinations nobody reads: over the non-combination nodes themselves (any number of terms: a sink)
        if (sinks_enabled) {
            std::vector<uint8_t> read(nodes.size(), 0);
            for (auto& nd : nodes) {
                if (nd.a >= 0) read[(size_t)nd.a] = 1;
                if (nd.b >= 0) read[(size_t)nd.b] = 1;
                if (nd.c >= 0) read[(size_t)nd.c] = 1;
                for (auto& t : nd.terms) read[(size_t)t.first] = 1;
            }
            for (size_t k = 0; k < N0; k++) {
                Node& nd = nodes[k];
                if (nd.opc != F_LIN || nd.hint == 0xffffffffu || read[k]) continue;
                std::map<int, long long> E;
                for (auto& t : nd.terms) E[t.first] += t.second;
                bool ok = true;
                for (int iter = 0; iter < 4096 && ok; iter++) {
                    int lin = -1;
                    for (auto& kv : E)
                        if (kv.second != 0 && nodes[(size_t)kv.first].opc == F_LIN && nodes[(size_t)kv.first].hint == 0xffffffffu) lin = kv.first;
                    if (lin < 0) break;   // (a hinted combination stays a term: its slot holds the value anyway)
                    const long long c = E[lin];
                    E.erase(lin);
                    for (auto& u : nodes[(size_t)lin].terms) E[u.first] += c * u.second;
                    if (E.size() > 255) ok = false;
                }
                std::vector<std::pair<int, int>> r;
                long long abs_sum = 0;
                for (auto& kv : E) {
                    if (kv.second == 0) continue;
                    if (std::llabs(kv.second) > F_MAX_COEF) ok = false;
                    abs_sum += std::llabs(kv.second);
                    r.push_back({kv.first, (int)kv.second});
                }
                // (h2e_field_sinks accumulates |coef| x value: the sum must stay below 2^12 w - what a record of 14 terms can reach)
                if (ok && r.size() <= 255 && abs_sum <= (long long)F_MAX_COEF * 14) nd.terms = r;
            }
        }
        // back into topological order by index (partial sums were appended behind their readers)
        {
            const size_t N = nodes.size();
            std::vector<int> order, new_of(N, -1);
            std::vector<uint8_t> state(N, 0);
            std::vector<std::pair<uint32_t, uint32_t>> st;   // (node, next dependency to visit)
            std::vector<std::vector<int>> deps(N);
            for (size_t k = 0; k < N; k++) {
                const Node& nd = nodes[k];
                if (nd.a >= 0) deps[k].push_back(nd.a);
                if (nd.b >= 0) deps[k].push_back(nd.b);
                if (nd.c >= 0) deps[k].push_back(nd.c);
                for (auto& t : nd.terms) deps[k].push_back(t.first);
            }
            for (size_t root = 0; root < N; root++) {
                if (state[root]) continue;
                st.push_back({(uint32_t)root, 0});
                state[root] = 1;
                while (!st.empty()) {
                    auto& top = st.back();
                    if (top.second < deps[top.first].size()) {
                        int x = deps[top.first][top.second++];
                        if (!state[(size_t)x]) {
                            state[(size_t)x] = 1;
                            st.push_back({(uint32_t)x, 0});
                        } else if (state[(size_t)x] == 1) {
                            throw std::runtime_error("field chain: cycle after rebalancing");
                        }
                    } else {
                        state[top.first] = 2;
                        new_of[top.first] = (int)order.size();
                        order.push_back((int)top.first);
                        st.pop_back();
                    }
                }
            }
            std::vector<Node> sorted(N);
            for (size_t q = 0; q < N; q++) {
                Node nd = nodes[(size_t)order[q]];
                if (nd.a >= 0) nd.a = new_of[(size_t)nd.a];
                if (nd.b >= 0) nd.b = new_of[(size_t)nd.b];
                if (nd.c >= 0) nd.c = new_of[(size_t)nd.c];
                for (auto& t : nd.terms) t.first = new_of[(size_t)t.first];
                std::sort(nd.terms.begin(), nd.terms.end());
                sorted[q] = std::move(nd);
            }
            nodes = std::move(sorted);
        }
    }

    // Which ops of the segment carry a hint the replay / expansion will ask for, and can all of them be predicted?
    // `check_only`: feasibility (before the compiler's dead-op pass), nothing is built.
    bool compile(FieldChain& out, bool check_only) {
        const size_t RW = digit_rows ? 16 : 8;
        out.rec_words = (uint32_t)RW;
        F_MAX_TERMS = (int)RW - 2;
        if (getenv("H2E_FIELD_TERMS")) F_MAX_TERMS = std::min(F_MAX_TERMS, atoi(getenv("H2E_FIELD_TERMS")));
        std::vector<uint8_t> needed(n_ops, 0);
        std::vector<uint32_t> stack;
        for (uint32_t i = 0; i < n_ops; i++) {
            const H2EOp& op = ops[i];
            if ((op.flags & H2E_FLAG_HINTED) && (op.opcode == H2E_OP_INT_MUL || op.opcode == H2E_OP_REDUCE || op.opcode == H2E_OP_DIV_CORE)) {
                if (op.flags & H2E_FLAG_HINT_STRIDED) { out.why = "strided hint"; return false; }
                needed[i] = 1;
                stack.push_back(i);
            }
        }
        if (aux)
            for (auto& kv : *aux)
                if (!needed[kv.first]) {
                    needed[kv.first] = 1;
                    stack.push_back(kv.first);
                }
        if (exports)
            for (auto& kv : *exports)
                if (!needed[kv.first]) {
                    needed[kv.first] = 1;
                    stack.push_back(kv.first);
                }
        if (stack.empty()) { out.why = "no hinted op"; return false; }
        std::map<uint32_t, int> import_index;   // first limb reference of an imported integer -> its virtual op index (n_ops + k)
        std::vector<Import> imports;
        while (!stack.empty()) {
            uint32_t i = stack.back();
            stack.pop_back();
            const H2EOp& op = ops[i];
            if (!is_int_op(op.opcode) && !is_cond_op(op.opcode)) { out.why = "op " + std::to_string(op.opcode) + " in the value cone"; return false; }
            if ((op.opcode == H2E_OP_ASSIGN_W || op.opcode == H2E_OP_ASSIGN_BIT || op.opcode == H2E_OP_ASSIGN) && (op.flags & H2E_FLAG_INPUT_STRIDED)) {
                out.why = "strided input";
                return false;
            }
            Opd o[3];
            int n = operands(op, o);
            for (int q = 0; q < n; q++) {
                int p = o[q].is_int ? int_producer(op.refs[o[q].refpos]) : cond_producer(op.refs[o[q].refpos]);
                if (p < 0 && o[q].is_int && import_of) {
                    Import imp;
                    uint32_t ref = op.refs[o[q].refpos];
                    if (import_of(ref, imp)) {
                        if (!import_index.count(ref)) {
                            import_index[ref] = (int)(n_ops + imports.size());
                            imports.push_back(imp);
                            if (note_import) note_import(ref);
                        }
                        continue;
                    }
                }
                if (p < 0 || (uint32_t)p >= i) { out.why = "operand of op " + std::to_string(i) + " (opcode " + std::to_string(op.opcode) + ") is not a result of this segment"; return false; }
                if (!needed[p]) {
                    needed[p] = 1;
                    stack.push_back((uint32_t)p);
                }
            }
        }
        if (check_only) return true;

        // ---- expressions / nodes, in program order -----------------------------------------------------------------
        expr.assign(n_ops + imports.size(), Expr());   // integer results (+ the imported integers, as virtual ops behind the segment's own)
        for (size_t k = 0; k < imports.size(); k++) {
            int m = new_node(imports[k].kind == 1 ? F_CONST_W : F_INPUT_W);
            nodes[m].imm = imports[k].imm;
            nodes[m].hint_src = imports[k].kind == 0;
            expr[n_ops + k] = single(m);
        }
        std::vector<int> cond(n_ops, -1);       // condition results: node
        for (uint32_t i = 0; i < n_ops; i++) {
            if (!needed[i]) continue;
            const H2EOp& op = ops[i];
            Opd o[3];
            int n = operands(op, o);
            int prod[3] = {-1, -1, -1};
            for (int q = 0; q < n; q++) {
                prod[q] = o[q].is_int ? int_producer(op.refs[o[q].refpos]) : cond_producer(op.refs[o[q].refpos]);
                if (prod[q] < 0 && o[q].is_int) {
                    auto it = import_index.find(op.refs[o[q].refpos]);
                    if (it != import_index.end()) prod[q] = it->second;
                }
            }
            const bool hinted = (op.flags & H2E_FLAG_HINTED) != 0;
            switch (op.opcode) {
                case H2E_OP_INT_MUL: {
                    int a = mat(prod[0]), b = mat(prod[1]);
                    int m = new_node(F_MUL);
                    nodes[m].a = a;
                    nodes[m].b = b;
                    if (hinted) nodes[m].hint = op.imm;
                    expr[i] = single(m);
                } break;
                case H2E_OP_DIV_CORE: {   // refs: b, a'
                    int b = mat(prod[0]), a = mat(prod[1]);
                    int m = new_node(F_DIV);
                    nodes[m].a = a;
                    nodes[m].b = b;
                    if (hinted) nodes[m].hint = op.imm;
                    expr[i] = single(m);
                } break;
                case H2E_OP_REDUCE: {
                    const Expr& e = expr[prod[0]];
                    if (!hinted) {
                        expr[i] = e;
                    } else if (e.t.size() == 1 && e.t[0].second == 1 && nodes[e.t[0].first].hint == op.imm) {
                        expr[i] = e;   // the reduce of a tagged value: the slot is already being filled
                    } else {
                        // a node of its own that owns the hint slot; consumers keep combining the operand's expression (as a
                        // value mod w the reduce is the identity), so the hint store is off the dependency path
                        int m = new_node(F_LIN);
                        nodes[m].terms = e.t;
                        nodes[m].hint = op.imm;
                        expr[i] = getenv("H2E_FIELD_REDUCE_NODES") ? single(m) : e;
                    }
                } break;
                case H2E_OP_INT_ADD: expr[i] = combine(prod[0], 1, prod[1], 1); break;
                case H2E_OP_INT_SUB: expr[i] = combine(prod[0], 1, prod[1], -1); break;
                case H2E_OP_INT_NEG: expr[i] = combine(prod[0], -1, -1, 1); break;
                case H2E_OP_INT_MUL_SMALL: expr[i] = combine(prod[0], (int)op.imm, -1, 1); break;
                case H2E_OP_MASK_INT: {   // a * coeff, coeff a condition
                    int m = new_node(F_SELECT);
                    nodes[m].c = cond[prod[1]];
                    nodes[m].a = mat(prod[0]);
                    nodes[m].b = -1;
                    expr[i] = single(m);
                } break;
                case H2E_OP_BISEC_INT: {
                    if (op.imm != 0 && (int)op.imm != L) { out.why = "bisec_int of another field"; return false; }
                    int m = new_node(F_SELECT);
                    nodes[m].c = cond[prod[0]];
                    nodes[m].a = mat(prod[1]);
                    nodes[m].b = mat(prod[2]);
                    expr[i] = single(m);
                } break;
                case H2E_OP_CONST_INT: {
                    int m = new_node(F_CONST_W);
                    nodes[m].imm = op.imm;
                    expr[i] = single(m);
                } break;
                case H2E_OP_CONST_INT_INPUT: case H2E_OP_ASSIGN_W: {
                    int m = new_node(F_INPUT_W);
                    nodes[m].imm = op.imm;
                    expr[i] = single(m);
                } break;
                case H2E_OP_IS_INT_ZERO: {
                    int m = new_node(F_ISZERO);
                    nodes[m].a = mat(prod[0]);
                    cond[i] = m;
                } break;
                case H2E_OP_NOT: {
                    int m = new_node(F_NOT);
                    nodes[m].a = cond[prod[0]];
                    cond[i] = m;
                } break;
                case H2E_OP_AND: case H2E_OP_OR: case H2E_OP_XNOR: {
                    int m = new_node(op.opcode == H2E_OP_AND ? F_AND : op.opcode == H2E_OP_OR ? F_OR : F_XNOR);
                    nodes[m].a = cond[prod[0]];
                    nodes[m].b = cond[prod[1]];
                    cond[i] = m;
                } break;
                case H2E_OP_ASSIGN_BIT: case H2E_OP_ASSIGN: {
                    int m = new_node(F_INPUT_FE);
                    nodes[m].imm = op.imm;
                    cond[i] = m;
                } break;
                case H2E_OP_CONST: {
                    int m = new_node(F_CONST_FE);
                    nodes[m].imm = op.imm;
                    cond[i] = m;
                } break;
                default: out.why = "unexpected op"; return false;
            }
            if (aux) {
                auto it = aux->find(i);
                if (it != aux->end()) {
                    int node = is_cond_op(op.opcode) ? cond[i] : (expr[i].t.size() == 1 && expr[i].t[0].second == 1 ? expr[i].t[0].first : -1);
                    if (node < 0 || nodes[node].hint != 0xffffffffu) { out.why = "aux hint on a node that cannot take it"; return false; }
                    nodes[node].hint = it->second;
                }
            }
        }
        if (exports)
            for (auto& kv : *exports) {
                if (!is_int_op(ops[kv.first].opcode)) { out.why = "a later segment reads a result that is not an integer"; return false; }
                int node = mat((int)kv.first);
                if (nodes[node].hint == kv.second) continue;
                if (nodes[node].hint != 0xffffffffu) {   // the node already fills another slot: a copy for this one
                    int m = new_node(F_LIN);
                    nodes[m].terms.push_back({node, 1});
                    node = m;
                }
                nodes[node].hint = kv.second;
            }
        auto deps_of = [&](const Node& nd, std::vector<uint32_t>& d) {
            d.clear();
            auto add = [&](int x) {
                if (x >= 0 && std::find(d.begin(), d.end(), (uint32_t)x) == d.end()) d.push_back((uint32_t)x);
            };
            add(nd.a);
            add(nd.b);
            add(nd.c);
            for (auto& kv : nd.terms) add(kv.first);
        };
        // ---- linear combinations of linear combinations ---------------------------------------------------------------
        // A LIN node exists where some consumer needed the value in a slot (an operand of a product, a hint); a later
        // combination that reads it pays a whole round for that.  Its terms are taken over instead - the deepest operands
        // first, for as long as the result fits a record: the same residue, one level less on the path (the inner node stays
        // for its other readers; recomputing is free here, rounds are not).
        if (!getenv("H2E_FIELD_NO_INLINE")) {
            std::vector<uint32_t> depth(nodes.size(), 0);
            std::vector<uint32_t> dp;
            for (size_t k = 0; k < nodes.size(); k++) {
                Node& nd = nodes[k];
                if (nd.opc == F_LIN) {
                    for (int iter = 0; iter < 64; iter++) {
                        uint32_t dm = 0;
                        for (auto& t : nd.terms) dm = std::max(dm, depth[t.first]);
                        if (dm <= 1) break;
                        std::map<int, long long> acc;
                        bool ok = true;
                        for (auto& t : nd.terms) {
                            if (depth[t.first] == dm) {
                                const Node& in = nodes[t.first];
                                if (in.opc != F_LIN) { ok = false; break; }
                                for (auto& u : in.terms) acc[u.first] += (long long)t.second * u.second;
                            } else {
                                acc[t.first] += t.second;
                            }
                        }
                        if (!ok) break;
                        std::vector<std::pair<int, int>> r;
                        for (auto& kv : acc) {
                            if (kv.second == 0) continue;
                            if (std::llabs(kv.second) > F_MAX_COEF) { ok = false; break; }
                            r.push_back({kv.first, (int)kv.second});
                        }
                        if (!ok || (int)r.size() > F_MAX_TERMS) break;
                        nd.terms = r;
                    }
                }
                dp.clear();
                deps_of(nd, dp);
                uint32_t dd = 0;
                for (uint32_t pp : dp) dd = std::max(dd, depth[pp]);
                depth[k] = dd + 1;
            }
        }
        // ---- depth-balanced combinations (round 5) -------------------------------------------------------------------------
        // combine() builds a sum the way the program adds it up: c = c + x, and every time the expression outgrows a record the
        // larger side is materialised - a 54-term coefficient of an Fq12 product becomes a CHAIN L14 -> L13 -> L13 -> L12, four
        // rounds deep, and the pass above cannot merge it (nothing fits).  The deepest path of a bn256 check's Miller loop was
        // 248 products and 774 combinations.  Here every combination is re-associated for depth: starting from its terms, the
        // deepest term is replaced by its own terms for as long as that lowers the depth the combination can be had at; what is
        // left is summed as a tree - the EARLIEST terms first, fourteen at a time, into partial sums that sit off the path (the
        // k-ary Huffman rule for the minimal maximum depth) - so the latest product is read by the final record directly.
        // Identical partial sums are shared.  A hinted combination nobody reads any more (the reduce of a value that is only
        // combined further: consumers flatten through it) is written over the products themselves and leaves the chain as a sink.
        // (only with the sinks kernel: inside the chain a hint-only combination waits for a free row, and while it waits it keeps the
        // products it now reads directly alive - 10 k value slots for a bn256 Miller loop, more than the LDS holds)
        long_lins = !getenv("H2E_FIELD_NO_LONG") && F_MAX_TERMS == 14;
        if (digit_rows && next_hint != nullptr && !getenv("H2E_FIELD_NO_SINKS") && !getenv("H2E_FIELD_NO_REBALANCE")) rebalance(true);
        // (Round 3 also tried products that compute their operands' combinations in their own row - two records, the second behind
        // the round's rows: 2 261 -> 1 538 rounds and a SLOWER chain, 2.9 -> 3.4 ms, a fused round being as long as the two it
        // replaces.  Taken out in round 5; the second-record mechanism now carries the long combinations.)
        // ---- hint-only combinations leave the chain -------------------------------------------------------------------
        // A quarter of the linear combinations feed nothing but their own hint slot (the reduce of a value nobody multiplies
        // again).  The chain is bound by its CU's instruction issue, and these are not on any path: they are computed after it
        // by a kernel of their own, one lane per (combination, instance), from the *finalized* values of their terms - the
        // combination is linear, so it holds for the canonical values as well as for the Montgomery residues.  A term that
        // has no hint slot of its own gets one (a store in the chain instead of a record).
        if (digit_rows && next_hint && !getenv("H2E_FIELD_NO_SINKS")) {
            std::vector<uint8_t> live(nodes.size(), 0), read(nodes.size(), 0);
            std::vector<uint32_t> st, dd;
            for (size_t k = 0; k < nodes.size(); k++)
                if (nodes[k].hint != 0xffffffffu) {
                    live[k] = 1;
                    st.push_back((uint32_t)k);
                }
            while (!st.empty()) {
                uint32_t k = st.back();
                st.pop_back();
                deps_of(nodes[k], dd);
                for (uint32_t x : dd) {
                    read[x] = 1;
                    if (!live[x]) {
                        live[x] = 1;
                        st.push_back(x);
                    }
                }
            }
            for (size_t k = 0; k < nodes.size(); k++) {
                Node& nd = nodes[k];
                if (!live[k] || read[k] || nd.opc != F_LIN || nd.hint == 0xffffffffu) continue;
                bool ok = nd.terms.size() <= 255;
                for (auto& t : nd.terms) {
                    const Node& in = nodes[t.first];
                    bool value_node = in.opc == F_LIN || in.opc == F_MUL || in.opc == F_DIV || in.opc == F_SELECT || in.opc == F_INPUT_W || in.opc == F_CONST_W;
                    if (!value_node || std::abs(t.second) > 255) ok = false;
                    if ((in.opc == F_INPUT_W || in.opc == F_CONST_W) && in.imm >= (1u << 21)) ok = false;
                }
                if (!ok) continue;
                out.sink_offsets.push_back((uint32_t)out.sink_words.size());
                out.sink_words.push_back(nd.hint);
                out.sink_words.push_back((uint32_t)nd.terms.size());
                for (auto& t : nd.terms) {
                    Node& in = nodes[t.first];
                    uint32_t kind, index;
                    if (in.opc == F_INPUT_W && in.hint_src) {   // a value an earlier segment left in a hint slot: canonical already
                        kind = 0;
                        index = in.imm;
                    } else if (in.opc == F_INPUT_W || in.opc == F_CONST_W) {
                        kind = in.opc == F_INPUT_W ? 1u : 2u;
                        index = in.imm;
                    } else {
                        if (in.hint == 0xffffffffu) in.hint = (*next_hint)++;
                        kind = 0;
                        index = in.hint;
                        if (index >= (1u << 21)) throw std::runtime_error("field chain: hint slot index beyond the sink records' 21 bits");
                    }
                    out.sink_words.push_back(((uint32_t)t.second & 0x1ffu) << 23 | kind << 21 | index);
                }
                nd.hint = 0xffffffffu;   // no longer a root: the node (and what only it needed) drops out of the chain
            }
        }
        // ---- alive nodes: what a hint-bearing node depends on --------------------------------------------------------
        const size_t N = nodes.size();
        std::vector<std::vector<uint32_t>> preds(N);
        std::vector<uint8_t> alive(N, 0);
        {
            std::vector<uint32_t> st, d;
            for (size_t k = 0; k < N; k++)
                if (nodes[k].hint != 0xffffffffu) {
                    alive[k] = 1;
                    st.push_back((uint32_t)k);
                }
            while (!st.empty()) {
                uint32_t k = st.back();
                st.pop_back();
                deps_of(nodes[k], d);
                for (uint32_t x : d)
                    if (!alive[x]) {
                        alive[x] = 1;
                        st.push_back(x);
                    }
            }
        }
        std::vector<std::vector<uint32_t>> succs(N);
        for (size_t k = 0; k < N; k++) {
            if (!alive[k]) continue;
            deps_of(nodes[k], preds[k]);
            for (uint32_t x : preds[k]) succs[x].push_back((uint32_t)k);
        }
        // ---- rounds: backwards, cheapest class first (see schedule_classes in h2e_capi.cpp), 64 records per round ---------
        // Digit rows: a record is a row of some wave and rows do not wait for each other inside a round, so a round may mix
        // products with light records - what matters is that the four rows of a WAVE are of one kind (a wave executes every kind
        // its rows hold, one after the other).  Rounds are therefore scheduled by dependency alone (their number drops from
        // class-pure 2 808 to 2 06x for bn256, the dependency depth being 1 950) and the emission below pads each kind to a
        // multiple of four rows; 54 rows (51 with fused products) leave room for the padding in a pass of 60.
        const bool mixed_rounds = digit_rows;   // (class-pure rounds - 2 808 instead of 2 06x for bn256 - were an A/B knob until round 5)
        // Two passes of the kernel's 60 rows per round (108 + padding; H2E_FIELD_STEP=<rows>: A/B): once the combinations are
        // depth-balanced the rounds are bound by their row capacity (bn256 Miller loop: depth 696, 50.7 k records = 940 rounds of 54,
        // 1 148 scheduled; 696 of 108) and a round costs ~1 340 cycles whatever is in it + ~31 per record: measured (64 x bn256, one
        // batch after the other) 54 rows: 4.04 ms, 80: 3.86, 108: 3.79, 160: 3.81
        const bool pair_products = digit_rows && w_words == 4 && !getenv("H2E_FIELD_NO_PAIRS");   // (eight-digit fields: two products per row, see the emission)
        size_t STEP = digit_rows ? 108 : 64;
        if (digit_rows && getenv("H2E_FIELD_STEP")) STEP = std::max<size_t>(8, std::min<size_t>(234, (size_t)atoi(getenv("H2E_FIELD_STEP"))));
        auto cls_of = [&](uint32_t k) -> int {   // 0 light, 1 loads, 2 products, 3 divisions
            switch (nodes[k].opc) {
                case F_MUL: return mixed_rounds ? 0 : 2;   // (digit rows: a product is a row like any other - see the round emission)
                case F_DIV: return 3;
                case F_INPUT_W: case F_INPUT_FE: case F_CONST_W: case F_CONST_FE: return 1;
                default: return 0;
            }
        };
        std::vector<uint8_t> is_sink(N, 0), done(N, 0);
        std::vector<uint32_t> left(N, 0);
        size_t n_left = 0;
        for (size_t k = 0; k < N; k++) {
            if (!alive[k]) continue;
            is_sink[k] = succs[k].empty() ? 1 : 0;
        }
        std::vector<uint32_t> ready[4], released;
        for (size_t k = 0; k < N; k++) {
            if (!alive[k] || is_sink[k]) continue;
            for (uint32_t s : succs[k])
                if (!is_sink[s]) left[k]++;
            n_left++;
        }
        for (size_t k = 0; k < N; k++)
            if (alive[k] && !is_sink[k] && left[k] == 0) ready[cls_of((uint32_t)k)].push_back((uint32_t)k);
        auto rows_of = [&](const std::vector<uint32_t>& rd) {
            size_t n_other = 0, n_mul = 0;
            for (uint32_t k : rd)
                if (pair_products && nodes[k].opc == F_MUL) n_mul++;
                else n_other++;
            return n_other + (n_mul + 1) / 2;
        };
        std::vector<std::vector<uint32_t>> rounds_rev;
        std::vector<int> rcls_rev;
        while (n_left > 0) {
            int c = -1;
            for (int q = 0; q < 4 && c < 0; q++)
                if (!ready[q].empty()) c = q;
            if (c < 0) throw std::runtime_error("field chain: scheduler stalled");
            std::vector<uint32_t> rd;
            // capacity in ROWS: a pair of products is one row
            size_t take = 0, n_other = 0, n_mul = 0;
            while (take < ready[c].size()) {
                const bool paired = pair_products && nodes[ready[c][ready[c].size() - 1 - take]].opc == F_MUL;
                const size_t o2 = n_other + (paired ? 0 : 1), m2 = n_mul + (paired ? 1 : 0);
                if (o2 + (m2 + 1) / 2 > STEP) break;
                n_other = o2;
                n_mul = m2;
                take++;
            }
            rd.assign(ready[c].end() - take, ready[c].end());
            ready[c].resize(ready[c].size() - take);
            for (uint32_t k : rd) {
                done[k] = 1;
                n_left--;
                for (uint32_t pp : preds[k])
                    if (!is_sink[pp] && --left[pp] == 0) released.push_back(pp);
            }
            for (uint32_t pp : released) ready[cls_of(pp)].push_back(pp);
            released.clear();
            rounds_rev.push_back(std::move(rd));
            rcls_rev.push_back(c);
        }
        std::vector<std::vector<uint32_t>> rounds(rounds_rev.rbegin(), rounds_rev.rend());
        std::vector<int> rcls(rcls_rev.rbegin(), rcls_rev.rend());
        std::vector<uint32_t> round_of(N, 0);
        for (size_t r = 0; r < rounds.size(); r++)
            for (uint32_t k : rounds[r]) round_of[k] = (uint32_t)r;
        // sinks (nodes that only feed a hint slot): the first round of their class after their operands with a free lane
        for (size_t k = 0; k < N; k++) {
            if (!alive[k] || !is_sink[k]) continue;
            size_t r0 = 0;
            for (uint32_t pp : preds[k]) {
                if (!done[pp]) throw std::runtime_error("field chain: sink reads an unplaced node");
                r0 = std::max<size_t>(r0, (size_t)round_of[pp] + 1);
            }
            int c = cls_of((uint32_t)k);
            bool placed = false;
            for (size_t r = r0; r < rounds.size() && !placed; r++)
                if (rcls[r] == c && rows_of(rounds[r]) + 1 <= STEP) {
                    rounds[r].push_back((uint32_t)k);
                    round_of[k] = (uint32_t)r;
                    placed = true;
                }
            if (!placed) {
                rounds.push_back({(uint32_t)k});
                rcls.push_back(c);
                round_of[k] = (uint32_t)rounds.size() - 1;
            }
            done[k] = 1;
        }
        // the rounds of loads (inputs, constants: no operands) go first: the kernel runs them in a loop of its own, so that
        // the main loop's body has no global load in it (a load anywhere in the body makes the compiler wait for the vector
        // memory counter - i.e. for the hint stores in flight - on every path)
        {
            std::vector<std::vector<uint32_t>> front, rest;
            std::vector<int> fcls, rcls2;
            for (size_t r = 0; r < rounds.size(); r++) {
                if (rcls[r] == 1) {
                    front.push_back(std::move(rounds[r]));
                    fcls.push_back(1);
                } else {
                    rest.push_back(std::move(rounds[r]));
                    rcls2.push_back(rcls[r]);
                }
            }
            out.n_load_rounds = (uint32_t)front.size();
            rounds = std::move(front);
            rounds.insert(rounds.end(), std::make_move_iterator(rest.begin()), std::make_move_iterator(rest.end()));
            rcls = fcls;
            rcls.insert(rcls.end(), rcls2.begin(), rcls2.end());
            for (size_t r = 0; r < rounds.size(); r++)
                for (uint32_t k : rounds[r]) round_of[k] = (uint32_t)r;
        }
        // ---- value slots by liveness over the rounds -------------------------------------------------------------------
        std::vector<uint32_t> last_use(N, 0);
        for (size_t k = 0; k < N; k++)
            if (alive[k])
                for (uint32_t pp : preds[k]) last_use[pp] = std::max(last_use[pp], round_of[k]);
        std::vector<int> slot(N, -1);
        std::vector<std::vector<int>> free_at(rounds.size() + 1);
        std::vector<int> free_list;
        int n_slots = 0;
        for (size_t r = 0; r < rounds.size(); r++) {
            for (int sl : free_at[r]) free_list.push_back(sl);
            for (uint32_t k : rounds[r]) {
                if (succs[k].empty() && nodes[k].opc != F_DIV) continue;   // only stored to its hint slot (a division's slot also carries the inverse from the loader wave to the rows)
                int sl;
                if (!free_list.empty()) {
                    sl = free_list.back();
                    free_list.pop_back();
                } else {
                    sl = n_slots++;
                }
                slot[k] = sl;
                free_at[std::min<size_t>((size_t)last_use[k] + 1, rounds.size())].push_back(sl);
            }
        }
        if (n_slots >= 0xfff0) { out.why = "too many value slots"; return false; }
        // the kernels keep their record chunks (a ring of H2E_DP_CHUNKS for the digit-row kernel, two for the lane kernel) and
        // every value slot in LDS (160 KB per workgroup on gfx950)
        if ((size_t)(digit_rows ? H2E_DP_CHUNKS : 2u) * H2E_WCHUNK * RW * 4 + (digit_rows ? H2E_DP_DIV_SCRATCH : 0u) + (size_t)n_slots * w_words * 8 + 64 > (size_t)160 * 1024) {
            out.why = "the value slots (" + std::to_string(n_slots) + ") do not fit the LDS";
            return false;
        }
        // ---- records -----------------------------------------------------------------------------------------------------
        auto pad_chunk = [&]() {
            while ((out.recs.size() / RW) % H2E_WCHUNK) out.recs.insert(out.recs.end(), RW, 0u);
        };
        auto slot_of = [&](int node) -> uint32_t {
            if (node < 0) return 0xffffu;
            if (slot[node] < 0) throw std::runtime_error("field chain: operand without a slot");
            return (uint32_t)slot[node];
        };
        // a round = one header record (word 0: count | kind << 8) followed by its records; a round never straddles a chunk:
        // a header with kind 0xff sends the kernel to the start of the next chunk (the rest of the chunk is padding).  The
        // headers travel through LDS with the records: a header read from global memory would make the computing wave wait
        // for every hint store it has in flight (loads and stores share one in-order counter).
        size_t term_hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // Eight-digit fields (bn256 Fq, bls12_381 Fr): a product uses half of its row's 16 lanes, so the products of a round go two to
        // a row (engine.hip DigitRow::mont_mul2): record = [MUL | 1 << 4 | dst << 16, hint, a | a2 << 16, b | b2 << 16, dst2, hint2].
        std::map<uint32_t, uint32_t> partner;   // first product of a paired row -> the second
        for (size_t r = 0; r < rounds.size(); r++) {
            auto& rd = rounds[r];
            // (a row of 16 lanes per record: the four records of a wave should be of one kind)
            const uint32_t PAD = 0xffffffffu;   // a padding row (NOP record)
            size_t n_conts = 0;                 // second records of long combinations: behind the round's rows
            if (digit_rows) {
                auto op_terms = [&](uint32_t k) { return nodes[k].terms.size(); };
                std::stable_sort(rd.begin(), rd.end(), [&](uint32_t x, uint32_t y) {
                    if (nodes[x].opc != nodes[y].opc) return nodes[x].opc < nodes[y].opc;
                    return op_terms(x) < op_terms(y);   // (a wave runs its longest combination's term loop)
                });
                if (mixed_rounds) {   // kinds: linear combinations | products | everything else, each from a wave boundary
                    auto kind_of = [&](uint32_t k) { return nodes[k].opc == F_LIN ? 0 : nodes[k].opc == F_MUL ? 2 : 3; };
                    std::vector<uint32_t> padded;
                    for (int kd = 0; kd < 4; kd++) {
                        size_t before = padded.size();
                        uint32_t open_pair = PAD;   // (eight-digit fields: two products per row - the second rides in the first's record)
                        for (uint32_t k : rd)
                            if (kind_of(k) == kd) {
                                if (kd == 2 && pair_products) {
                                    if (open_pair == PAD) {
                                        open_pair = k;
                                        padded.push_back(k);
                                    } else {
                                        partner[open_pair] = k;
                                        open_pair = PAD;
                                    }
                                    continue;
                                }
                                padded.push_back(k);
                            }
                        if (padded.size() > before)
                            while (padded.size() % 4) padded.push_back(PAD);
                    }
                    while (!padded.empty() && padded.back() == PAD) padded.pop_back();
                    if (padded.size() > (STEP <= 54 ? 60 : 250)) throw std::runtime_error("field chain: a padded round exceeds a pass of the kernel");
                    rd = padded;
                }
                for (uint32_t k : rd)
                    if (k != PAD && nodes[k].opc == F_LIN && (int)nodes[k].terms.size() > F_MAX_TERMS) n_conts++;   // a long combination's second record
                if (rd.size() + n_conts > 254) throw std::runtime_error("field chain: a round's records exceed the 8-bit record index");
            }
            size_t at = out.recs.size() / RW;
            if (at % H2E_WCHUNK + 1 + rd.size() + n_conts > H2E_WCHUNK) {
                uint32_t padh[16] = {0xff00u};
                out.recs.insert(out.recs.end(), padh, padh + RW);
                pad_chunk();
            }
            at = out.recs.size() / RW;
            out.rounds.push_back((uint32_t)at);
            out.rounds.push_back((uint32_t)rd.size() | ((uint32_t)rcls[r] << 8));
            {
                uint32_t max_terms = 0;   // of the round's linear combinations (the kernel's term loop runs that far)
                for (uint32_t k : rd)
                    if (k != PAD && nodes[k].opc == F_LIN) max_terms = std::max<uint32_t>(max_terms, (uint32_t)nodes[k].terms.size());
                term_hist[std::min<uint32_t>(max_terms <= 6 ? max_terms : 7, 7)]++;
                // (word 1: the lane kernel's longest combination of the round / the digit kernel's count of second records behind the rows)
                uint32_t hdr[16] = {(uint32_t)rd.size() | ((uint32_t)rcls[r] << 8), digit_rows ? (uint32_t)n_conts : max_terms};
                out.recs.insert(out.recs.end(), hdr, hdr + RW);
            }
            std::vector<uint32_t> conts;
            for (uint32_t k : rd) {
                uint32_t w[16] = {0};
                if (k == PAD) {
                    w[0] = F_NOP | 0xffffu << 16;
                    out.recs.insert(out.recs.end(), w, w + RW);
                    continue;
                }
                const Node& nd = nodes[k];
                // digit rows: word 0 = opcode (4 bits) | terms of the combination (4 bits; a product: 1 = a pair of products) | index of
                // the second record of a long combination, counted from the round's first record (8 bits; 0 = none) | destination slot
                auto head = [&](uint32_t opc, uint32_t nt, uint32_t cidx, uint32_t dst) { return opc | nt << 4 | cidx << 8 | dst << 16; };
                w[0] = digit_rows ? head(nd.opc, (uint32_t)std::min<size_t>(nd.terms.size(), 14), 0, slot[k] >= 0 ? (uint32_t)slot[k] : 0xffffu)
                                  : nd.opc | ((slot[k] >= 0 ? (uint32_t)slot[k] : 0xffffu) << 16);
                w[1] = nd.hint == 0xffffffffu ? 0u : nd.hint + 1;
                if (digit_rows) {   // 16-word records: the hint slot in 18 bits, the sum of a combination's coefficients above it
                    if (w[1] >= (1u << 18)) throw std::runtime_error("field chain: hint slot index beyond the records' 18 bits");
                    int sum = 0;
                    if (nd.opc == F_LIN)
                        for (auto& t : nd.terms) sum += t.second;
                    w[1] |= (uint32_t)sum << 18;   // (|sum| <= 14 x 255: 13 bits and a sign)
                }
                if (nd.hint != 0xffffffffu) {
                    out.note_hint(nd.hint, hint_split);
                }
                switch (nd.opc) {
                    case F_LIN:
                        if ((int)nd.terms.size() > (digit_rows ? 2 : 1) * F_MAX_TERMS) throw std::runtime_error("field chain: LIN with too many terms");
                        for (size_t t = 0; t + 2 < RW; t++)
                            w[2 + t] = t < nd.terms.size() ? (slot_of(nd.terms[t].first) | ((uint32_t)(uint16_t)(int16_t)nd.terms[t].second << 16)) : 0u;   // unused: coefficient 0 (slot 0)
                        if ((int)nd.terms.size() > F_MAX_TERMS) {   // a long combination: the other terms in a second record behind the round's rows
                            uint32_t w2[16] = {0};
                            const size_t rest = nd.terms.size() - (size_t)F_MAX_TERMS;
                            for (size_t t = 0; t < rest; t++)
                                w2[2 + t] = slot_of(nd.terms[(size_t)F_MAX_TERMS + t].first) | ((uint32_t)(uint16_t)(int16_t)nd.terms[(size_t)F_MAX_TERMS + t].second << 16);
                            w2[0] = head(F_CONT, (uint32_t)rest, 0, 0xffffu);
                            const uint32_t cidx = (uint32_t)(rd.size() + conts.size() / RW);
                            if (cidx == 0 || cidx > 255) throw std::runtime_error("field chain: second record of a long combination out of reach");
                            w[0] |= cidx << 8;
                            conts.insert(conts.end(), w2, w2 + RW);
                        }
                        out.n_lin++;
                        break;
                    case F_MUL: case F_DIV: case F_AND: case F_OR: case F_XNOR:
                        w[2] = slot_of(nd.a);
                        w[3] = slot_of(nd.b);
                        if (nd.opc == F_MUL) out.n_mul++;
                        if (nd.opc == F_MUL && partner.count(k)) {   // the row's second product
                            const uint32_t k2 = partner[k];
                            const Node& n2 = nodes[k2];
                            w[0] |= 1u << 4;
                            w[2] |= slot_of(n2.a) << 16;
                            w[3] |= slot_of(n2.b) << 16;
                            w[4] = slot[k2] >= 0 ? (uint32_t)slot[k2] : 0xffffu;
                            w[5] = n2.hint == 0xffffffffu ? 0u : n2.hint + 1;
                            if (w[5] >= (1u << 18)) throw std::runtime_error("field chain: hint slot index beyond the records' 18 bits");
                            if (n2.hint != 0xffffffffu) out.note_hint(n2.hint, hint_split);
                            out.n_mul++;
                        }
                        break;
                    case F_ISZERO: case F_NOT: w[2] = slot_of(nd.a); break;
                    case F_SELECT:
                        w[2] = slot_of(nd.c);
                        w[3] = slot_of(nd.a);
                        w[4] = slot_of(nd.b);
                        break;
                    default:
                        w[2] = nd.imm;
                        if (nd.hint_src) w[0] |= 0x100u;   // H2E_F_INPUT_W from the hint workspace (tape.h)
                        break;
                }
                out.recs.insert(out.recs.end(), w, w + RW);
            }
            if (conts.size() != n_conts * RW) throw std::runtime_error("field chain: second records miscounted");
            out.recs.insert(out.recs.end(), conts.begin(), conts.end());
        }
        pad_chunk();
        if (getenv("H2E_FIELD_STATS")) {
            {   // the dependency depth: what a schedule with mixed rounds could get down to
                std::vector<uint32_t> depth(N, 0), depth_mul(N, 0);
                uint32_t dmax = 0, dmul = 0;
                for (size_t k = 0; k < N; k++) {
                    if (!alive[k]) continue;
                    uint32_t dd = 0, dm = 0;
                    for (uint32_t pp : preds[k]) {
                        dd = std::max(dd, depth[pp]);
                        dm = std::max(dm, depth_mul[pp]);
                    }
                    depth[k] = dd + 1;
                    depth_mul[k] = dm + (nodes[k].opc == F_MUL ? 1 : 0);
                    dmax = std::max(dmax, depth[k]);
                    dmul = std::max(dmul, depth_mul[k]);
                }
                fprintf(stderr, "field chain: dependency depth %u nodes (%u products on the deepest product path)\n", dmax, dmul);
                if (getenv("H2E_FIELD_PATH")) {   // the deepest path, last node first: opcode / terms
                    size_t at = 0;
                    for (size_t k = 0; k < N; k++)
                        if (alive[k] && depth[k] == dmax) at = k;
                    std::string line;
                    size_t hist[16] = {0};
                    while (true) {
                        hist[nodes[at].opc & 15]++;
                        char b[32];
                        if (nodes[at].opc == F_LIN) snprintf(b, sizeof b, "L%zu ", nodes[at].terms.size());
                        else snprintf(b, sizeof b, "%c ", "?LMDZNAOXS"[nodes[at].opc < 10 ? nodes[at].opc : 0]);
                        line += b;
                        size_t nxt = N;
                        for (uint32_t pp : preds[at])
                            if (depth[pp] + 1 == depth[at]) nxt = pp;
                        if (nxt == N) break;
                        at = nxt;
                    }
                    fprintf(stderr, "%s\n", line.substr(0, 6000).c_str());
                    for (int q = 0; q < 14; q++) fprintf(stderr, "opc %d: %zu  ", q, hist[q]);
                    fprintf(stderr, "\n");
                }
            }
            {   // sinks: records that only feed a hint slot
                size_t n_sink = 0, n_sink_lin = 0, n_sink_easy = 0, n_lin = 0, sink_terms = 0;
                for (size_t k = 0; k < N; k++) {
                    if (!alive[k]) continue;
                    if (nodes[k].opc == F_LIN) n_lin++;
                    if (!is_sink[k]) continue;
                    n_sink++;
                    if (nodes[k].opc != F_LIN) continue;
                    n_sink_lin++;
                    sink_terms += nodes[k].terms.size();
                    bool easy = true;
                    for (auto& t : nodes[k].terms) {
                        const Node& in = nodes[t.first];
                        if (!(in.hint != 0xffffffffu || in.opc == F_INPUT_W || in.opc == F_CONST_W)) easy = false;
                    }
                    n_sink_easy += easy;
                }
                size_t th[16] = {0};
                for (size_t k = 0; k < N; k++)
                    if (alive[k] && nodes[k].opc == F_LIN) th[std::min<size_t>(nodes[k].terms.size(), 15)]++;
                fprintf(stderr, "field chain: linear combinations by terms:");
                for (int q = 0; q < 16; q++) fprintf(stderr, " %zu", th[q]);
                fprintf(stderr, "\n");
                fprintf(stderr, "field chain: %zu linear combinations, %zu sinks (%zu linear combinations, %zu terms; %zu of them over hinted values / inputs only)\n", n_lin, n_sink,
                        n_sink_lin, sink_terms, n_sink_easy);
            }
            {   // long combinations: how many, and in how many rounds (a round is as long as its longest row)
                size_t n_long = 0, rounds_with = 0;
                for (size_t r = 0; r < rounds.size(); r++) {
                    bool any = false;
                    for (uint32_t k : rounds[r])
                        if (k != 0xffffffffu && nodes[k].opc == F_LIN && (int)nodes[k].terms.size() > F_MAX_TERMS) {
                            n_long++;
                            any = true;
                        }
                    rounds_with += any;
                }
                fprintf(stderr, "field chain: %zu long combinations in %zu of %zu rounds\n", n_long, rounds_with, rounds.size());
            }
            size_t cnt[4] = {0, 0, 0, 0}, ops_in[4] = {0, 0, 0, 0};
            for (size_t r = 0; r < rounds.size(); r++) {
                cnt[rcls[r]]++;
                ops_in[rcls[r]] += rounds[r].size();
            }
            fprintf(stderr, "field chain rounds: light %zu (%zu ops), loads %zu, products %zu (%zu ops), divisions %zu\n", cnt[0], ops_in[0], cnt[1], cnt[2], ops_in[2], cnt[3]);
            fprintf(stderr, "   %d value slots = %zu KB of LDS next to %zu KB of record chunks; %zu hint-only combinations computed after the chain\n", n_slots,
                    (size_t)n_slots * w_words * 8 / 1024, (size_t)2 * H2E_WCHUNK * RW * 4 / 1024, out.sink_offsets.size());
            fprintf(stderr, "   rounds by their longest linear combination (0 .. 6 terms, more): %zu %zu %zu %zu %zu %zu %zu, %zu\n", term_hist[0], term_hist[1],
                    term_hist[2], term_hist[3], term_hist[4], term_hist[5], term_hist[6], term_hist[7]);
        }
        out.n_slots = (uint32_t)std::max(1, n_slots);
        out.n_nodes = 0;
        for (size_t k = 0; k < N; k++) out.n_nodes += alive[k];
        return true;
    }
};



// ================================================================================================
// Hint store: what the values-only replay of a segment with field hints shrinks to.
//
// Once every mul-like result has its canonical value in a hint slot, every value the full expansion needs in place
// before it starts - the results that escape their sub-range (tape.h H2E_FLAG_LOCAL_RESULT) - is a function of hints,
// constants and inputs alone: a mul-like result *is* its hint, and a light result (int_add / int_sub / int_neg /
// int_mul_small_constant chains, src/circuit/integer_chip.rs:384-464, :618-658) is a limb-wise integer combination
//     limb_i = sum_j coef_j * limb_i(leaf_j) + K_i,       native = sum_j coef_j * native(leaf_j) + K_native  (mod n)
// of the mul-like results / constants / inputs it was built from, K being the (signed) sum of the
// find_w_modulus_of_ceil_times constants its int_sub / int_neg steps added - known when the program is compiled.  No
// stored value depends on another stored value, so there is no chain left: one lane per (stored op, instance), instances
// minor (coalesced like the expansion), engine.hip h2e_hint_store.
//
// Store op record (32-bit words):  w0 = kind | n_terms << 8 | K index << 16,  w1 = base row,  w2 = range row,
//   w3.. = terms: leaf kind (bits 30-31: 0 hint slot, 1 pool word offset, 2 input slot) | (coef + 128) << 22 | index (22 bits)
// kinds: S_W      mul-like result cells (limbs: range column 0 of rows w2 + 3 i, native: base (w1, 4)) <- its one leaf
//        S_LIN    add-like result cells (limb i: base (w1 + i, 4), native: base (w1 + L, 4)) <- the combination
//        S_FE     a condition cell base (w1, 4) <- raw word of hint slot (term 0)
//        S_CONST  assign_int_constant rows (limb i: base (w1 + i, 0), native (w1 + L, 0)) <- pool constant (term 0)
//        S_FULL   the tape op w1 (segment relative) run as it is (assign / assign_w / constants made from inputs)
struct HintStore {
    std::vector<uint32_t> words;      // the records
    std::vector<uint32_t> offsets;    // per store op: first word of its record
    std::vector<uint64_t> ktab;       // K constants: (2 L + 4) words each (limbs: 2 words each, native: 4 words)
    std::map<uint32_t, uint32_t> aux_hint;   // op index -> hint slot the field chain must fill (conditions, masked integers)
    std::vector<uint32_t> ext;               // extension leaves (tape.h H2EStoreExt), H2E_SX_WORDS words each
    std::map<std::array<uint32_t, H2E_SX_WORDS>, uint32_t> ext_index;
    uint32_t n_terms_max = 0;
    std::string why;
};

struct StoreCompiler {
    const H2EOp* ops;
    uint32_t n_ops;
    int L;
    const H2EFieldConsts* fc;
    const FieldCompiler* fcmp;      // operand resolution (shares producer / bounds)
    uint32_t next_aux;              // first free hint slot

    struct Lin {
        std::map<uint32_t, int> leaf;      // leaf word (kind << 30 | index) -> coef
        std::map<uint32_t, int> ceil;      // times -> multiplicity of ceil constant C_times
    };
    static void add_scaled(Lin& r, const Lin& a, int s) {
        for (auto& kv : a.leaf) {
            int& c = r.leaf[kv.first];
            c += s * kv.second;
            if (c == 0) r.leaf.erase(kv.first);
        }
        for (auto& kv : a.ceil) {
            int& c = r.ceil[kv.first];
            c += s * kv.second;
            if (c == 0) r.ceil.erase(kv.first);
        }
    }
    static bool supported(uint16_t oc) {
        switch (oc) {
            case H2E_OP_NOP: case H2E_OP_ASSIGN_W: case H2E_OP_ASSIGN: case H2E_OP_ASSIGN_BIT: case H2E_OP_CONST_INT: case H2E_OP_CONST_INT_INPUT:
            case H2E_OP_CONST: case H2E_OP_INT_ADD: case H2E_OP_INT_SUB: case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: case H2E_OP_INT_MUL:
            case H2E_OP_REDUCE: case H2E_OP_IS_INT_ZERO: case H2E_OP_NOT: case H2E_OP_MASK_INT: case H2E_OP_DIV_CORE: case H2E_OP_SUM_LIMBS:
            case H2E_OP_ASSERT_CONST: case H2E_OP_AND: case H2E_OP_OR: case H2E_OP_XNOR:
                return true;
            default: return false;
        }
    }
    // can every op of the segment be handled (before the dead-op pass: any of them may turn out to be stored)?
    bool feasible(std::string& why) const {
        for (uint32_t i = 0; i < n_ops; i++) {
            const H2EOp& op = ops[i];
            if (!supported(op.opcode)) { why = "hint store: opcode " + std::to_string(op.opcode); return false; }
            if ((op.opcode == H2E_OP_ASSIGN_W || op.opcode == H2E_OP_ASSIGN || op.opcode == H2E_OP_ASSIGN_BIT) && (op.flags & H2E_FLAG_INPUT_STRIDED)) {
                why = "hint store: strided input";
                return false;
            }
            if (op.opcode == H2E_OP_INT_ADD || op.opcode == H2E_OP_INT_SUB || op.opcode == H2E_OP_INT_NEG || op.opcode == H2E_OP_INT_MUL_SMALL) {
                FieldCompiler::Opd o[3];
                int n = fcmp->operands(op, o);
                for (int q = 0; q < n; q++)
                    if (fcmp->int_producer(op.refs[o[q].refpos]) < 0) {
                        FieldCompiler::Import imp;   // (a value of an earlier segment of the same context: read from where its chain left it)
                        if (fcmp->import_of && fcmp->import_of(op.refs[o[q].refpos], imp)) continue;
                        why = "hint store: operand of a light op from outside the segment";
                        return false;
                    }
            }
        }
        return true;
    }
    std::vector<Lin> lin;
    std::vector<uint8_t> have;
    uint32_t aux_of(HintStore& out, uint32_t i) {
        auto it = out.aux_hint.find(i);
        if (it != out.aux_hint.end()) return it->second;
        uint32_t s = next_aux++;
        out.aux_hint[i] = s;
        return s;
    }
    // the integer result of op p as a combination of leaves
    const Lin& flatten(HintStore& out, uint32_t p) {
        if (have[p]) return lin[p];
        const H2EOp& op = ops[p];
        Lin r;
        auto leaf = [&](uint32_t kind, uint32_t index) {
            if (index >= (1u << 22)) throw std::runtime_error("hint store: leaf index out of range");
            r.leaf[(kind << 30) | index] = 1;
        };
        FieldCompiler::Opd o[3];
        int n = fcmp->operands(op, o);
        int prod[3] = {-1, -1, -1};
        for (int q = 0; q < n; q++)
            if (o[q].is_int) prod[q] = fcmp->int_producer(op.refs[o[q].refpos]);
        // the integer operand q as a combination: a result of this segment, or - a context cut into several segments - an integer of
        // an earlier one, read from its CELLS (below)
        Lin imported[3];
        auto operand = [&](int q) -> const Lin& {   // (a reference: copying an operand's map per use was most of a pairing shape's 3.6 s of compile time)
            if (prod[q] >= 0) return flatten(out, (uint32_t)prod[q]);
            // an integer of an earlier segment of the same context: read as its cells hold it (that segment's own store - or its
            // expansion's inputs before it - put them in place before this segment's value chain starts).  Not through a hint slot:
            // the limbs of an unreduced value are not the canonical split of its residue.
            FieldCompiler::Import imp;
            uint32_t ref = op.refs[o[q].refpos];
            if (!fcmp->import_of || !fcmp->import_of(ref, imp)) throw std::runtime_error("hint store: operand from outside the segment");
            std::array<uint32_t, H2E_SX_WORDS> e{};
            e[0] = H2E_SX_CELLS;
            for (int i = 0; i <= L; i++) e[1 + i] = op.refs[o[q].refpos + i];
            uint32_t idx;
            auto it = out.ext_index.find(e);
            if (it != out.ext_index.end()) idx = it->second;
            else {
                idx = (uint32_t)(out.ext.size() / H2E_SX_WORDS);
                out.ext.insert(out.ext.end(), e.begin(), e.end());
                out.ext_index[e] = idx;
            }
            if (idx >= (1u << 22)) throw std::runtime_error("hint store: leaf index out of range");
            Lin& x = imported[q];
            x.leaf.clear();
            x.leaf[(3u << 30) | idx] = 1;
            return x;
        };
        switch (op.opcode) {
            case H2E_OP_INT_MUL: case H2E_OP_REDUCE: case H2E_OP_DIV_CORE:
                if (!(op.flags & H2E_FLAG_HINTED)) throw std::runtime_error("hint store: a live mul-like result without a hint");
                leaf(0, op.imm);
                break;
            case H2E_OP_MASK_INT: leaf(0, aux_of(out, p)); break;
            case H2E_OP_CONST_INT: leaf(1, op.imm); break;
            case H2E_OP_ASSIGN_W: case H2E_OP_CONST_INT_INPUT: leaf(2, op.imm); break;
            case H2E_OP_INT_ADD:
                add_scaled(r, operand(0), 1);
                add_scaled(r, operand(1), 1);
                break;
            case H2E_OP_INT_SUB:   // a - b + C_(b.times)   (integer_chip.rs:408-437)
                add_scaled(r, operand(0), 1);
                add_scaled(r, operand(1), -1);
                r.ceil[op.imm] += 1;
                if (r.ceil[op.imm] == 0) r.ceil.erase(op.imm);
                break;
            case H2E_OP_INT_NEG:   // C_(a.times) - a       (:439-464)
                add_scaled(r, operand(0), -1);
                r.ceil[op.imm] += 1;
                if (r.ceil[op.imm] == 0) r.ceil.erase(op.imm);
                break;
            case H2E_OP_INT_MUL_SMALL:
                add_scaled(r, operand(0), (int)op.imm);
                break;
            default: throw std::runtime_error("hint store: not an integer result");
        }
        lin[p] = std::move(r);
        have[p] = 1;
        return lin[p];
    }
    std::map<std::vector<uint64_t>, uint32_t> k_index;
    std::map<std::map<uint32_t, int>, uint32_t> k_memo;   // the ceil multiplicities of a combination -> its K table entry (a few dozen
                                                          // distinct ones per segment; the big-number arithmetic below once for each)
    uint32_t k_of(HintStore& out, const Lin& e) {
        auto memo = k_memo.find(e.ceil);
        if (memo != k_memo.end()) return memo->second;
        uint32_t idx = k_build(out, e);
        k_memo[e.ceil] = idx;
        return idx;
    }
    uint32_t k_build(HintStore& out, const Lin& e) {
        // K = sum mult_t * C_t: limbs modulo 2^128 (the true limb values are non-negative and below 2^128), native modulo n
        std::vector<uint64_t> k((size_t)2 * L + 4, 0);
        for (int i = 0; i < L; i++) {
            unsigned __int128 acc = 0;
            for (auto& kv : e.ceil) {
                unsigned __int128 c = ((unsigned __int128)fc->ceil_limbs[kv.first][i][1] << 64) | fc->ceil_limbs[kv.first][i][0];
                acc += (unsigned __int128)(__int128)kv.second * c;   // two's complement: a negative multiplicity wraps
            }
            k[2 * i] = (uint64_t)acc;
            k[2 * i + 1] = (uint64_t)(acc >> 64);
        }
        HBig n = HBig::from_words(fc->n, 4), pos, neg;
        for (auto& kv : e.ceil) {
            HBig c = HBig::from_words(fc->ceil_native[kv.first], 4);
            if (kv.second > 0) pos = pos + c * HBig((uint64_t)kv.second);
            else neg = neg + c * HBig((uint64_t)(-kv.second));
        }
        HBig nat = (pos + n * (neg / n + HBig(1)) - neg) % n;
        nat.to_words(&k[(size_t)2 * L], 4);
        auto it = k_index.find(k);
        if (it != k_index.end()) return it->second;
        uint32_t idx = (uint32_t)(out.ktab.size() / ((size_t)2 * L + 4));
        out.ktab.insert(out.ktab.end(), k.begin(), k.end());
        k_index[k] = idx;
        return idx;
    }
    bool compile(HintStore& out) {
        lin.assign(n_ops, Lin());
        have.assign(n_ops, 0);
        {   // entry 0 of the K table: zero
            Lin zero;
            k_of(out, zero);
        }
        auto emit = [&](uint32_t kind, uint32_t k_idx, uint32_t w1, uint32_t w2, const std::vector<uint32_t>& terms) {
            if (terms.size() > 255 || k_idx > 0xffff) throw std::runtime_error("hint store: record field overflow");
            out.offsets.push_back((uint32_t)out.words.size());
            out.words.push_back(kind | ((uint32_t)terms.size() << 8) | (k_idx << 16));
            out.words.push_back(w1);
            out.words.push_back(w2);
            out.words.insert(out.words.end(), terms.begin(), terms.end());
            out.n_terms_max = std::max<uint32_t>(out.n_terms_max, (uint32_t)terms.size());
        };
        auto term = [](uint32_t leaf_word, int coef) -> uint32_t {
            if (coef < -127 || coef > 127) throw std::runtime_error("hint store: coefficient out of range");
            return (leaf_word & 0xc0000000u) | ((uint32_t)(coef + 128) << 22) | (leaf_word & 0x3fffffu);
        };
        for (uint32_t i = 0; i < n_ops; i++) {
            const H2EOp& op = ops[i];
            if (op.flags & H2E_FLAG_VALUES_SKIP) continue;
            bool local = (op.flags & H2E_FLAG_LOCAL_RESULT) != 0;
            switch (op.opcode) {
                case H2E_OP_NOP: case H2E_OP_ASSERT_CONST: case H2E_OP_SUM_LIMBS: break;
                case H2E_OP_ASSIGN_W: case H2E_OP_ASSIGN: case H2E_OP_ASSIGN_BIT: case H2E_OP_CONST: case H2E_OP_CONST_INT_INPUT:
                    emit(H2E_S_FULL, 0, i, 0, {});
                    break;
                case H2E_OP_CONST_INT: emit(H2E_S_CONST, 0, op.base_row, 0, {term((1u << 30) | op.imm, 1)}); break;
                case H2E_OP_INT_MUL: case H2E_OP_REDUCE: case H2E_OP_DIV_CORE:
                    if (local) break;
                    if (!(op.flags & H2E_FLAG_HINTED)) { out.why = "hint store: stored mul-like result without a hint"; return false; }
                    emit(H2E_S_W, 0, op.base_row, op.range_row, {term(op.imm, 1)});
                    break;
                case H2E_OP_MASK_INT:
                    if (local) break;
                    emit(H2E_S_LIN, 0, op.base_row, 0, {term(aux_of(out, i), 1)});
                    break;
                case H2E_OP_INT_ADD: case H2E_OP_INT_SUB: case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: {
                    if (local) break;
                    const Lin& e = flatten(out, i);
                    std::vector<uint32_t> terms;
                    long weight = 1;   // the native sum is below (1 + sum |coef|) n; the kernel's quotient estimate (h2e_hint_store, H2E_S_LIN)
                                       // takes one 64-bit word off its top: good below 2^12 n
                    for (auto& kv : e.leaf) {
                        terms.push_back(term(kv.first, kv.second));
                        weight += std::abs(kv.second);
                    }
                    if (weight >= 4096) { out.why = "hint store: a combination's coefficients sum to 4096 or more"; return false; }
                    emit(H2E_S_LIN, k_of(out, e), op.base_row, 0, terms);
                } break;
                case H2E_OP_IS_INT_ZERO: case H2E_OP_NOT:
                    if (local) break;
                    emit(H2E_S_FE, 0, fcmp->fe_row(op), 0, {term(aux_of(out, i), 1)});
                    break;
                case H2E_OP_AND: case H2E_OP_OR: case H2E_OP_XNOR:
                    emit(H2E_S_FE, 0, op.base_row, 0, {term(aux_of(out, i), 1)});
                    break;
                default: out.why = "hint store: opcode " + std::to_string(op.opcode); return false;
            }
        }
        // masked integers that only feed other values still need their aux hint: flatten() registered them
        return true;
    }
};

}  // namespace h2e
