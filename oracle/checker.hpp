// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Not part of the product path.
//
// Constraint checker: the MockProver-equivalent for exactly the gates / lookups / copy constraints
// the reference configures (this is what every reference test asserts, src/tests/mod.rs:117-132):
//   base gate            src/circuit/base_chip.rs:50-69
//   range lookups+gates  src/circuit/range_chip.rs:118-220, table :230-258
//   select lookup_any    src/circuit/select_chip.rs:71-88
//   copy constraints     src/context.rs:523-541
// Unassigned advice / fixed cells evaluate to zero, as in halo2's MockProver.
#pragma once
#include <map>
#include <array>
#include <string>
#include "records.hpp"

namespace h2o {

struct CheckReport {
    size_t base_gate_failures = 0, range_gate_failures = 0, range_lookup_failures = 0;
    size_t select_lookup_failures = 0, permutation_failures = 0;
    std::string first_error;
    bool ok() const {
        return base_gate_failures + range_gate_failures + range_lookup_failures + select_lookup_failures +
                   permutation_failures ==
               0;
    }
    void note(const std::string& s) {
        if (first_error.empty()) first_error = s;
    }
};

inline CheckReport check_records(const Records& rec) {
    CheckReport rep;
    const RecordsInner& in = *rec.inner;
    auto badv = [&](size_t row, int col) -> Fr {
        if (row >= in.base_rows) return Fr::zero();
        const AdvCell& c = in.base_adv[row * VAR_COLUMNS + col];
        return c.present ? c.val : Fr::zero();
    };
    auto bfix = [&](size_t row, int col) -> Fr {
        if (row >= in.base_rows) return Fr::zero();
        const FixCell& c = in.base_fix[row * FIXED_COLUMNS + col];
        return c.present ? c.val : Fr::zero();
    };
    // base gate
    for (size_t r = 0; r < rec.base_height; r++) {
        Fr acc = bfix(r, VAR_COLUMNS + MUL_COLUMNS + 1) + badv(r + 1, VAR_COLUMNS - 1) * bfix(r, VAR_COLUMNS + MUL_COLUMNS);
        for (int i = 0; i < VAR_COLUMNS; i++) acc = acc + badv(r, i) * bfix(r, i);
        for (int i = 0; i < MUL_COLUMNS; i++) acc = acc + badv(r, 2 * i) * badv(r, 2 * i + 1) * bfix(r, VAR_COLUMNS + i);
        if (!acc.is_zero()) {
            rep.base_gate_failures++;
            rep.note("base gate fails at row " + std::to_string(r));
        }
    }
    // range chip
    auto radv = [&](size_t row, int col) -> Fr {
        if (row >= in.range_rows) return Fr::zero();
        const AdvCell& c = in.range_adv[row * RANGE_ADV_COLUMNS + col];
        return c.present ? c.val : Fr::zero();
    };
    auto rfix = [&](size_t row, int col) -> Fr {
        if (row >= in.range_rows) return Fr::zero();
        const FixCell& c = in.range_fix[row * RANGE_FIX_COLUMNS + col];
        return c.present ? c.val : Fr::zero();
    };
    Fr shift_unit = Fr::from_u64(1ull << COMMON_RANGE_BITS);
    Fr f1 = Fr::one(), f2 = Fr::from_u64(2), f3 = Fr::from_u64(3);
    for (size_t r = 0; r < rec.range_height + 1; r++) {
        // lookups
        {
            BigUint tag = rfix(r, TagCol).to_bn();
            BigUint v = radv(r, TaggedRangeCol).to_bn();
            bool ok = tag <= BigUint(COMMON_RANGE_BITS) && v < (BigUint(1) << tag.low_u64());
            BigUint cv = radv(r, CommonRangeCol).to_bn();
            bool ok2 = cv < (BigUint(1) << COMMON_RANGE_BITS);
            if (!ok || !ok2) {
                rep.range_lookup_failures++;
                rep.note("range lookup fails at row " + std::to_string(r));
            }
        }
        Fr lines = rfix(r, AccLinesCol);
        Fr acc_v = radv(r, ValueAccCol);
        // one line
        {
            Fr acc = (acc_v - radv(r, TaggedRangeCol)) * lines * (lines - f2) * (lines - f3);
            if (!acc.is_zero()) {
                rep.range_gate_failures++;
                rep.note("range 1-line gate fails at row " + std::to_string(r));
            }
        }
        for (int nl = 2; nl <= 3; nl++) {
            Fr acc = acc_v;
            Fr shift = Fr::one();
            for (int j = 0; j < nl; j++) {
                acc = acc - radv(r + j, CommonRangeCol) * shift;
                shift = shift * shift_unit;
            }
            for (int j = 0; j < nl; j++) {
                acc = acc - radv(r + j, TaggedRangeCol) * shift;
                shift = shift * shift_unit;
            }
            acc = acc * lines;
            for (int root = 1; root <= 3; root++)
                if (root != nl) acc = acc * (lines - (root == 1 ? f1 : root == 2 ? f2 : f3));
            if (!acc.is_zero()) {
                rep.range_gate_failures++;
                rep.note("range " + std::to_string(nl) + "-line gate fails at row " + std::to_string(r));
            }
        }
    }
    // select chip lookup_any
    {
        typedef std::array<uint64_t, 8> Key;
        std::map<Key, int> table;
        auto sadv = [&](size_t row, int col) -> Fr {
            if (row >= in.select_rows) return Fr::zero();
            const AdvCell& c = in.select_adv[row * 2 + col];
            return c.present ? c.val : Fr::zero();
        };
        auto sfix = [&](size_t row, int col) -> Fr {
            if (row >= in.select_rows) return Fr::zero();
            const FixCell& c = in.select_fix[row * 2 + col];
            return c.present ? c.val : Fr::zero();
        };
        auto key = [&](const Fr& a, const Fr& b) {
            Key k;
            a.to_canonical(&k[0]);
            b.to_canonical(&k[4]);
            return k;
        };
        for (size_t r = 0; r < rec.select_height + 1; r++)
            if (sfix(r, SelIsLookupCol).is_zero()) table[key(sadv(r, SelValueCol), sfix(r, SelEncodeCol))] = 1;
        table[key(Fr::zero(), Fr::zero())] = 1;  // unused rows of the 2^k-row circuit
        Fr shift = Fr::from_bn(BigUint(1) << 128);
        for (size_t r = 0; r < rec.select_height + 1; r++) {
            Fr enc = sadv(r, SelSelectCol) * shift + sfix(r, SelEncodeCol);
            if (!table.count(key(sadv(r, SelValueCol), enc))) {
                rep.select_lookup_failures++;
                rep.note("select lookup fails at row " + std::to_string(r));
            }
        }
    }
    // permutations
    for (auto& p : rec.permutations) {
        auto get = [&](const Cell& c, bool& ok) -> Fr {
            const AdvCell* a = nullptr;
            if (c.region == BaseChip && c.row < in.base_rows) a = &in.base_adv[(size_t)c.row * VAR_COLUMNS + c.col];
            if (c.region == RangeChip && c.row < in.range_rows) a = &in.range_adv[(size_t)c.row * RANGE_ADV_COLUMNS + c.col];
            if (c.region == SelectChip && c.row < in.select_rows) a = &in.select_adv[(size_t)c.row * 2 + c.col];
            ok = a && a->present && a->permute;
            return a ? a->val : Fr::zero();
        };
        bool ok1, ok2;
        Fr a = get(p.first, ok1), b = get(p.second, ok2);
        if (!ok1 || !ok2 || a != b) {
            rep.permutation_failures++;
            rep.note("permutation fails: (" + std::to_string(p.first.region) + "," + std::to_string(p.first.col) + "," +
                     std::to_string(p.first.row) + ") vs (" + std::to_string(p.second.region) + "," +
                     std::to_string(p.second.col) + "," + std::to_string(p.second.row) + ")");
        }
    }
    return rep;
}

}  // namespace h2o
