// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Not part of the product path.
//
// L0 record store + L1 row primitives, restating:
//   src/assign.rs:5-29,84-85,123-162   (Chip, Cell, AssignedValue, AssignedCondition, ValueSchema/pair!)
//   src/context.rs:36-46,135-158       (Context, clone_without_permutation)
//   src/context.rs:241-301             (RecordsInner, Records)
//   src/context.rs:590-997             (enable_permute, one_line, one_line_with_last, select + range writers)
//   src/circuit/base_chip.rs:81-605    (BaseChipOps and its impl on Context)
// The halo2 binding (`_assign_to_*`, `assign_all*`, context.rs:310-588) is out of scope.
#pragma once
#include <memory>
#include <vector>
#include <utility>
#include "field.hpp"

namespace h2o {

enum Chip : uint8_t { BaseChip = 0, RangeChip = 1, SelectChip = 2 };  // assign.rs:5-10

struct Cell {  // assign.rs:12-23
    uint8_t region;
    uint8_t col;
    uint32_t row;
    Cell() : region(0), col(0), row(0) {}
    Cell(Chip r, int c, size_t rw) : region(r), col((uint8_t)c), row((uint32_t)rw) {}
    bool operator==(const Cell& o) const { return region == o.region && col == o.col && row == o.row; }
};

struct AssignedValue {  // assign.rs:25-29
    Cell cell;
    Fr val;
    AssignedValue() : val(Fr::zero()) {}
    AssignedValue(Chip region, int col, size_t row, const Fr& v) : cell(region, col, row), val(v) {}
};

struct AssignedCondition {  // assign.rs:84-85
    AssignedValue v;
    AssignedCondition() {}
    explicit AssignedCondition(const AssignedValue& a) : v(a) {}
};

// assign.rs:123-162 — an operand is an already assigned cell or a bare value
struct ValueSchema {
    bool assigned;
    Cell cell;
    Fr val;
    ValueSchema(const Fr& v) : assigned(false), val(v) {}
    ValueSchema(const AssignedValue& a) : assigned(true), cell(a.cell), val(a.val) {}
    ValueSchema(const AssignedValue* a) : assigned(true), cell(a->cell), val(a->val) {}
};
typedef std::pair<ValueSchema, Fr> Pair;  // pair!(x, y)
inline Pair pr(const AssignedValue& a, const Fr& c) { return Pair(ValueSchema(a), c); }
inline Pair pr(const Fr& v, const Fr& c) { return Pair(ValueSchema(v), c); }

static const int VAR_COLUMNS = 5;                                  // base_chip.rs:14
static const int MUL_COLUMNS = 2;                                  // base_chip.rs:15
static const int FIXED_COLUMNS = VAR_COLUMNS + MUL_COLUMNS + 2;    // base_chip.rs:16
static const int RANGE_ADV_COLUMNS = 3;                            // range_chip.rs:28
static const int RANGE_FIX_COLUMNS = 2;                            // range_chip.rs:29
static const uint64_t COMMON_RANGE_BITS = 18;                      // range_chip.rs:22-24
static const uint64_t MAX_CHUNKS = 3;                              // range_chip.rs:22
static const uint64_t RANGE_VALUE_DECOMPOSE = 6;                   // range_chip.rs:31
static const uint64_t OVERFLOW_BITS = 6;                           // context.rs:38
enum RangeAdvCol { ValueAccCol = 0, TaggedRangeCol = 1, CommonRangeCol = 2 };  // range_chip.rs:81-86
enum RangeFixCol { AccLinesCol = 0, TagCol = 1 };                              // range_chip.rs:88-92
enum SelectAdvCol { SelValueCol = 0, SelSelectCol = 1 };                       // select_chip.rs:42-46
enum SelectFixCol { SelEncodeCol = 0, SelIsLookupCol = 1 };                    // select_chip.rs:48-52

struct AdvCell {  // (Option<N>, bool)
    Fr val;
    uint8_t present;
    uint8_t permute;
};
struct FixCell {  // Option<N>
    Fr val;
    uint8_t present;
};

// context.rs:241-252.  The reference pre-allocates MAX_ROWS = 1<<23 rows (context.rs:36,254-292);
// the oracle grows on demand (single-threaded use) or is pre-sized with reserve_rows().
struct RecordsInner {
    std::vector<AdvCell> base_adv;    // [row][5]
    std::vector<FixCell> base_fix;    // [row][9]
    std::vector<AdvCell> range_adv;   // [row][3]
    std::vector<FixCell> range_fix;   // [row][2]
    std::vector<AdvCell> select_adv;  // [row][2]
    std::vector<FixCell> select_fix;  // [row][2]
    size_t base_rows = 0, range_rows = 0, select_rows = 0;
    bool frozen = false;  // set while several threads write disjoint rows (no growth allowed)

    static AdvCell adv0() {
        AdvCell c;
        c.val = Fr::zero();
        c.present = 0;
        c.permute = 0;
        return c;
    }
    static FixCell fix0() {
        FixCell c;
        c.val = Fr::zero();
        c.present = 0;
        return c;
    }
    void ensure_base(size_t row) {
        if (row < base_rows) return;
        if (frozen) throw std::runtime_error("records frozen: base rows not reserved");
        size_t n = std::max<size_t>(1024, base_rows * 2);
        while (n <= row) n *= 2;
        base_adv.resize(n * VAR_COLUMNS, adv0());
        base_fix.resize(n * FIXED_COLUMNS, fix0());
        base_rows = n;
    }
    void ensure_range(size_t row) {
        if (row < range_rows) return;
        if (frozen) throw std::runtime_error("records frozen: range rows not reserved");
        size_t n = std::max<size_t>(1024, range_rows * 2);
        while (n <= row) n *= 2;
        range_adv.resize(n * RANGE_ADV_COLUMNS, adv0());
        range_fix.resize(n * RANGE_FIX_COLUMNS, fix0());
        range_rows = n;
    }
    void ensure_select(size_t row) {
        if (row < select_rows) return;
        if (frozen) throw std::runtime_error("records frozen: select rows not reserved");
        size_t n = std::max<size_t>(1024, select_rows * 2);
        while (n <= row) n *= 2;
        select_adv.resize(n * 2, adv0());
        select_fix.resize(n * 2, fix0());
        select_rows = n;
    }
    void reserve_rows(size_t b, size_t r, size_t s) {
        bool f = frozen;
        frozen = false;
        if (b) ensure_base(b);
        if (r) ensure_range(r);
        if (s) ensure_select(s);
        frozen = f;
    }
    AdvCell& adv(const Cell& c) {
        switch (c.region) {
            case BaseChip: ensure_base(c.row); return base_adv[(size_t)c.row * VAR_COLUMNS + c.col];
            case RangeChip: ensure_range(c.row); return range_adv[(size_t)c.row * RANGE_ADV_COLUMNS + c.col];
            default: ensure_select(c.row); return select_adv[(size_t)c.row * 2 + c.col];
        }
    }
};

typedef std::pair<Cell, Cell> Permutation;

// context.rs:294-301 + writers context.rs:590-997
struct Records {
    std::shared_ptr<RecordsInner> inner;
    size_t base_height = 0, range_height = 0, select_height = 0;
    std::vector<Permutation> permutations;

    Records() : inner(std::make_shared<RecordsInner>()) {}

    void enable_permute(const Cell& cell) { inner->adv(cell).permute = 1; }  // context.rs:590-608

    void assign_adv_cell_in_base_chip(size_t offset, int col, const Fr& val) {  // context.rs:610-620
        inner->ensure_base(offset);
        AdvCell& c = inner->base_adv[offset * VAR_COLUMNS + col];
        c.val = val;
        c.present = 1;
    }
    void assign_fix_cell_in_base_chip(size_t offset, int col, const Fr& val) {  // context.rs:622-632
        inner->ensure_base(offset);
        FixCell& c = inner->base_fix[offset * FIXED_COLUMNS + col];
        c.val = val;
        c.present = 1;
    }

    // context.rs:634-683
    void one_line(size_t offset, const std::vector<Pair>& base_coeff_pairs, const Fr* constant,
                  const std::vector<Fr>& mul_coeffs, const Fr* next) {
        assert(base_coeff_pairs.size() <= (size_t)VAR_COLUMNS);
        if (offset >= base_height) base_height = offset + 1;
        inner->ensure_base(offset);
        for (size_t i = 0; i < base_coeff_pairs.size(); i++) {
            const ValueSchema& base = base_coeff_pairs[i].first;
            if (base.assigned) {
                Cell new_cell(BaseChip, (int)i, offset);
                enable_permute(new_cell);
                enable_permute(base.cell);
                permutations.push_back(Permutation(base.cell, new_cell));
            }
            assign_adv_cell_in_base_chip(offset, (int)i, base.val);
            assign_fix_cell_in_base_chip(offset, (int)i, base_coeff_pairs[i].second);
        }
        for (size_t i = 0; i < mul_coeffs.size(); i++)
            assign_fix_cell_in_base_chip(offset, VAR_COLUMNS + (int)i, mul_coeffs[i]);
        if (next) {
            assign_fix_cell_in_base_chip(offset, VAR_COLUMNS + MUL_COLUMNS, *next);
        } else {
            assert(!inner->base_fix[offset * FIXED_COLUMNS + VAR_COLUMNS + MUL_COLUMNS].present);
        }
        if (constant) {
            assign_fix_cell_in_base_chip(offset, VAR_COLUMNS + MUL_COLUMNS + 1, *constant);
        } else {
            assert(!inner->base_fix[offset * FIXED_COLUMNS + VAR_COLUMNS + MUL_COLUMNS + 1].present);
        }
    }

    // context.rs:685-714
    void one_line_with_last(size_t offset, const std::vector<Pair>& base_coeff_pairs, const Pair& tail,
                            const Fr* constant, const std::vector<Fr>& mul_coeffs, const Fr* next) {
        assert(base_coeff_pairs.size() <= (size_t)VAR_COLUMNS - 1);
        one_line(offset, base_coeff_pairs, constant, mul_coeffs, next);
        int i = VAR_COLUMNS - 1;
        if (tail.first.assigned) {
            Cell new_cell(BaseChip, i, offset);
            enable_permute(new_cell);
            enable_permute(tail.first.cell);
            permutations.push_back(Permutation(tail.first.cell, new_cell));
        }
        assign_adv_cell_in_base_chip(offset, i, tail.first.val);
        assign_fix_cell_in_base_chip(offset, i, tail.second);
    }

    void ensure_range_record_size(size_t offset) {  // context.rs:716-720 (quirk Q4: height = last row + 2)
        if (offset >= range_height) range_height = offset + 1;
    }

    void assign_adv_cell_in_select_chip(size_t offset, int col, const Fr& val) {  // context.rs:722-735
        inner->ensure_select(offset);
        AdvCell& c = inner->select_adv[offset * 2 + col];
        c.val = val;
        c.present = 1;
    }
    void assign_fix_cell_in_select_chip(size_t offset, int col, const Fr& val) {  // context.rs:737-747
        inner->ensure_select(offset);
        FixCell& c = inner->select_fix[offset * 2 + col];
        c.val = val;
        c.present = 1;
    }

    // context.rs:749-767
    void assign_cache_value(size_t offset, const AssignedValue& v, const Fr& encode) {
        if (offset >= select_height) select_height = offset + 1;
        assign_adv_cell_in_select_chip(offset, SelValueCol, v.val);
        Cell idx(SelectChip, SelValueCol, offset);
        permutations.push_back(Permutation(idx, v.cell));
        enable_permute(idx);
        enable_permute(v.cell);
        assign_fix_cell_in_select_chip(offset, SelEncodeCol, encode);
        assign_fix_cell_in_select_chip(offset, SelIsLookupCol, Fr::zero());
    }

    // context.rs:769-801
    AssignedValue assign_select_value(size_t offset, const AssignedValue& v, const Fr& encode,
                                      const AssignedValue& selector) {
        if (offset >= select_height) select_height = offset + 1;
        assign_adv_cell_in_select_chip(offset, SelValueCol, v.val);
        assign_adv_cell_in_select_chip(offset, SelSelectCol, selector.val);
        Cell selector_cell(SelectChip, SelSelectCol, offset);
        permutations.push_back(Permutation(selector_cell, selector.cell));
        enable_permute(selector_cell);
        enable_permute(selector.cell);
        assign_fix_cell_in_select_chip(offset, SelEncodeCol, encode);
        assign_fix_cell_in_select_chip(offset, SelIsLookupCol, Fr::one());
        return AssignedValue(SelectChip, SelValueCol, offset, v.val);
    }

    void assign_adv_cell_in_range_chip(size_t offset, int col, const Fr& val) {  // context.rs:803-815
        inner->ensure_range(offset);
        AdvCell& c = inner->range_adv[offset * RANGE_ADV_COLUMNS + col];
        c.val = val;
        c.present = 1;
    }
    void assign_fix_cell_in_range_chip(size_t offset, int col, const Fr& val) {  // context.rs:817-833
        inner->ensure_range(offset);
        FixCell& c = inner->range_fix[offset * RANGE_FIX_COLUMNS + col];
        c.val = val;
        c.present = 1;
    }

    // context.rs:835-857
    AssignedValue assign_one_line_range_value(size_t offset, const Fr* v, const Fr& v_acc, uint64_t bits) {
        assert(bits <= COMMON_RANGE_BITS);
        ensure_range_record_size(offset + 1);
        assign_fix_cell_in_range_chip(offset, AccLinesCol, Fr::one());
        assign_fix_cell_in_range_chip(offset, TagCol, Fr::from_u64(bits));
        assign_adv_cell_in_range_chip(offset, TaggedRangeCol, v[0]);
        assign_adv_cell_in_range_chip(offset, ValueAccCol, v_acc);
        return AssignedValue(RangeChip, ValueAccCol, offset, v_acc);
    }

    // context.rs:859-907
    AssignedValue assign_two_line_range_value(size_t offset, const Fr* v, const Fr& v_acc, uint64_t bits) {
        assert(bits >= COMMON_RANGE_BITS * 2);
        assert(bits <= COMMON_RANGE_BITS * 4);
        ensure_range_record_size(offset + 2);
        assign_fix_cell_in_range_chip(offset, AccLinesCol, Fr::one() + Fr::one());
        assign_adv_cell_in_range_chip(offset, CommonRangeCol, v[0]);
        assign_adv_cell_in_range_chip(offset + 1, CommonRangeCol, v[1]);
        uint64_t cell_bits = (bits >= 3 * COMMON_RANGE_BITS) ? COMMON_RANGE_BITS : bits % COMMON_RANGE_BITS;
        assign_fix_cell_in_range_chip(offset, TagCol, Fr::from_u64(cell_bits));
        assign_adv_cell_in_range_chip(offset, TaggedRangeCol, v[2]);
        cell_bits = (bits > 3 * COMMON_RANGE_BITS) ? bits - 3 * COMMON_RANGE_BITS : 0;
        assign_fix_cell_in_range_chip(offset + 1, TagCol, Fr::from_u64(cell_bits));
        assign_adv_cell_in_range_chip(offset + 1, TaggedRangeCol, v[3]);
        assign_adv_cell_in_range_chip(offset, ValueAccCol, v_acc);
        return AssignedValue(RangeChip, ValueAccCol, offset, v_acc);
    }

    // context.rs:909-972
    AssignedValue assign_three_line_range_value(size_t offset, const Fr* v, const Fr& v_acc, uint64_t bits) {
        assert(bits >= COMMON_RANGE_BITS * 3);
        assert(bits <= COMMON_RANGE_BITS * 6);
        ensure_range_record_size(offset + 3);
        assign_fix_cell_in_range_chip(offset, AccLinesCol, Fr::one() + Fr::one() + Fr::one());
        assign_adv_cell_in_range_chip(offset, CommonRangeCol, v[0]);
        assign_adv_cell_in_range_chip(offset + 1, CommonRangeCol, v[1]);
        assign_adv_cell_in_range_chip(offset + 2, CommonRangeCol, v[2]);
        uint64_t cell_bits = (bits >= 4 * COMMON_RANGE_BITS) ? COMMON_RANGE_BITS : bits % COMMON_RANGE_BITS;
        assign_fix_cell_in_range_chip(offset, TagCol, Fr::from_u64(cell_bits));
        assign_adv_cell_in_range_chip(offset, TaggedRangeCol, v[3]);
        if (bits >= 5 * COMMON_RANGE_BITS)
            cell_bits = COMMON_RANGE_BITS;
        else if (bits > 4 * COMMON_RANGE_BITS)
            cell_bits = bits % COMMON_RANGE_BITS;
        else
            cell_bits = 0;
        assign_fix_cell_in_range_chip(offset + 1, TagCol, Fr::from_u64(cell_bits));
        assign_adv_cell_in_range_chip(offset + 1, TaggedRangeCol, v[4]);
        cell_bits = (bits > 5 * COMMON_RANGE_BITS) ? bits - 5 * COMMON_RANGE_BITS : 0;
        assign_fix_cell_in_range_chip(offset + 2, TagCol, Fr::from_u64(cell_bits));
        assign_adv_cell_in_range_chip(offset + 2, TaggedRangeCol, v[5]);
        assign_adv_cell_in_range_chip(offset, ValueAccCol, v_acc);
        return AssignedValue(RangeChip, ValueAccCol, offset, v_acc);
    }

    // context.rs:974-997; returns (cell, rows consumed)
    std::pair<AssignedValue, size_t> assign_range_value(size_t offset, std::vector<Fr> v, const Fr& v_acc,
                                                        uint64_t bits) {
        if (bits <= COMMON_RANGE_BITS) {
            return std::make_pair(assign_one_line_range_value(offset, v.data(), v_acc, bits), (size_t)1);
        } else if (bits < 2 * COMMON_RANGE_BITS) {
            throw std::runtime_error("unreachable: range bits in (18,36)");
        } else if (bits <= 4 * COMMON_RANGE_BITS) {
            v.resize(4, Fr::zero());
            return std::make_pair(assign_two_line_range_value(offset, v.data(), v_acc, bits), (size_t)2);
        } else if (bits <= 6 * COMMON_RANGE_BITS) {
            v.resize(6, Fr::zero());
            return std::make_pair(assign_three_line_range_value(offset, v.data(), v_acc, bits), (size_t)3);
        }
        throw std::runtime_error("unreachable: range bits > 108");
    }
};

struct PanicError : std::runtime_error {  // a reference-side assert!/unwrap panic
    explicit PanicError(const std::string& s) : std::runtime_error(s) {}
};

// context.rs:40-46 with BaseChipOps (base_chip.rs:81-605) implemented on it.
struct Context {
    Records records;
    size_t base_offset = 0, range_offset = 0, select_offset = 0;

    Context clone_without_permutation() const {  // context.rs:145-158
        Context c;
        c.records.inner = records.inner;
        c.records.base_height = records.base_height;
        c.records.range_height = records.range_height;
        c.records.select_height = records.select_height;
        c.base_offset = base_offset;
        c.range_offset = range_offset;
        c.select_offset = select_offset;
        return c;
    }

    // ---- BaseChipOps ----
    int var_columns() const { return VAR_COLUMNS; }
    int mul_columns() const { return MUL_COLUMNS; }
    void enable_permute(const AssignedValue& x) { records.enable_permute(x.cell); }

    // base_chip.rs:516-539
    std::vector<AssignedValue> one_line(const std::vector<Pair>& pairs, const Fr* constant,
                                        const std::vector<Fr>& mul_coeffs, const Fr* next) {
        std::vector<AssignedValue> res;
        for (size_t i = 0; i < pairs.size(); i++)
            res.push_back(AssignedValue(BaseChip, (int)i, base_offset, pairs[i].first.val));
        records.one_line(base_offset, pairs, constant, mul_coeffs, next);
        base_offset += 1;
        return res;
    }
    std::vector<AssignedValue> one_line_add(const std::vector<Pair>& pairs, const Fr* constant) {  // :94-100
        return one_line(pairs, constant, std::vector<Fr>(), nullptr);
    }
    // base_chip.rs:541-572
    std::pair<std::vector<AssignedValue>, AssignedValue> one_line_with_last(
        const std::vector<Pair>& pairs, const Pair& last, const Fr* constant, const std::vector<Fr>& mul_coeffs,
        const Fr* next) {
        std::vector<AssignedValue> res0;
        for (size_t i = 0; i < pairs.size(); i++)
            res0.push_back(AssignedValue(BaseChip, (int)i, base_offset, pairs[i].first.val));
        AssignedValue res1(BaseChip, VAR_COLUMNS - 1, base_offset, last.first.val);
        records.one_line_with_last(base_offset, pairs, last, constant, mul_coeffs, next);
        base_offset += 1;
        return std::make_pair(res0, res1);
    }

    typedef std::pair<const AssignedValue*, Fr> Elem;

    // base_chip.rs:110-132
    AssignedValue sum_with_constant_in_one_line(const std::vector<Elem>& elems, const Fr* constant) {
        assert((int)elems.size() < var_columns());
        Fr sum = Fr::zero();
        bool first = true;
        for (auto& e : elems) {
            Fr t = e.first->val * e.second;
            sum = first ? t : sum + t;
            first = false;
        }
        assert(!first);
        if (constant) sum = *constant + sum;
        std::vector<Pair> pairs;
        for (auto& e : elems) pairs.push_back(pr(*e.first, e.second));
        auto cells = one_line_with_last(pairs, pr(sum, -Fr::one()), constant, std::vector<Fr>(), nullptr);
        return cells.second;
    }
    // base_chip.rs:134-153
    AssignedValue sum_with_constant(const std::vector<Elem>& elems, const Fr* constant) {
        size_t columns = (size_t)var_columns();
        if (elems.size() < columns) return sum_with_constant_in_one_line(elems, constant);
        std::vector<Elem> curr(elems.begin(), elems.begin() + (columns - 1));
        AssignedValue acc = sum_with_constant_in_one_line(curr, constant);
        size_t pos = columns - 1;
        while (pos < elems.size()) {
            size_t end = std::min(elems.size(), pos + (columns - 2));
            std::vector<Elem> chunk(elems.begin() + pos, elems.begin() + end);
            AssignedValue acc_copy = acc;
            chunk.push_back(Elem(&acc_copy, Fr::one()));
            acc = sum_with_constant_in_one_line(chunk, nullptr);
            pos = end;
        }
        return acc;
    }
    AssignedValue add(const AssignedValue& a, const AssignedValue& b) {  // :155-160
        return sum_with_constant({Elem(&a, Fr::one()), Elem(&b, Fr::one())}, nullptr);
    }
    AssignedValue add_constant(const AssignedValue& a, const Fr& c) {  // :162-167
        return sum_with_constant({Elem(&a, Fr::one())}, &c);
    }
    AssignedValue mul(const AssignedValue& a, const AssignedValue& b) {  // :176-193
        Fr one = Fr::one(), zero = Fr::zero();
        Fr c = a.val * b.val;
        auto cells = one_line_with_last({pr(a, zero), pr(b, zero)}, pr(c, -one), nullptr, {one}, nullptr);
        return cells.second;
    }
    // base_chip.rs:219-243
    AssignedValue mul_add(const AssignedValue& a, const AssignedValue& b, const Fr& ab_coeff, const AssignedValue& c,
                          const Fr& c_coeff) {
        Fr one = Fr::one(), zero = Fr::zero();
        Fr d = a.val * b.val * ab_coeff + c.val * c_coeff;
        auto cells =
            one_line_with_last({pr(a, zero), pr(b, zero), pr(c, c_coeff)}, pr(d, -one), nullptr, {ab_coeff}, nullptr);
        return cells.second;
    }
    struct MulAddTerm {
        const AssignedValue *a, *b, *c;
        Fr c_coeff;
    };
    // base_chip.rs:245-281
    AssignedValue mul_add_with_next_line(const std::vector<MulAddTerm>& ls) {
        assert(ls.size() > 0);
        if (ls.size() == 1) return mul_add(*ls[0].a, *ls[0].b, Fr::one(), *ls[0].c, ls[0].c_coeff);
        Fr one = Fr::one(), zero = Fr::zero();
        Fr t = zero;
        Fr neg_one = -one;
        for (size_t i = 0; i < ls.size(); i++) {
            const MulAddTerm& l = ls[i];
            one_line_with_last({pr(*l.a, zero), pr(*l.b, zero), pr(*l.c, l.c_coeff)},
                               i == 0 ? pr(t, zero) : pr(t, one), nullptr, {one}, &neg_one);
            t = l.a->val * l.b->val + l.c->val * l.c_coeff + t;
        }
        auto cells = one_line_with_last({}, pr(t, zero), nullptr, {}, nullptr);
        return cells.second;
    }
    // base_chip.rs:298-321
    std::pair<AssignedCondition, AssignedValue> invert(const AssignedValue& a) {
        Fr zero = Fr::zero(), one = Fr::one();
        Fr b = a.val.inv_or_zero();
        Fr c = one - a.val * b;
        auto cells = one_line({pr(a, zero), pr(c, zero)}, nullptr, {one}, nullptr);
        AssignedValue cc = cells[1];
        Fr neg_one = -one;
        auto cells2 = one_line_with_last({pr(a, zero), pr(b, zero)}, pr(cc, one), &neg_one, {one}, nullptr);
        return std::make_pair(AssignedCondition(cells2.second), cells2.first[1]);
    }
    AssignedCondition is_zero(const AssignedValue& a) { return invert(a).first; }  // :323-325
    AssignedValue assign_constant(const Fr& v) {  // :344-349
        auto cells = one_line_add({pr(v, -Fr::one())}, &v);
        return cells[0];
    }
    AssignedValue assign(const Fr& v) {  // :351-355
        auto cells = one_line_add({pr(v, Fr::zero())}, nullptr);
        return cells[0];
    }
    AssignedCondition assign_bit(const Fr& a) {  // :357-367 (quirk Q2: two unconstrained copies)
        Fr zero = Fr::zero(), one = Fr::one();
        auto cells = one_line({pr(a, one), pr(a, zero)}, nullptr, {-one}, nullptr);
        return AssignedCondition(cells[0]);
    }
    void assert_constant(const AssignedValue& a, const Fr& b) {  // :375-379
        if (!(a.val == b)) throw PanicError("assert_constant: assert_eq!(a.val, b) failed");
        one_line_add({pr(a, -Fr::one())}, &b);
    }
    void assert_bit(const AssignedValue& a) {  // :381-390
        Fr zero = Fr::zero(), one = Fr::one();
        one_line({pr(a, one), pr(a, zero)}, nullptr, {-one}, nullptr);
    }
    AssignedCondition and_(const AssignedCondition& a, const AssignedCondition& b) {  // :392-396
        return AssignedCondition(mul(a.v, b.v));
    }
    AssignedCondition not_(const AssignedCondition& a) {  // :398-403
        Fr one = Fr::one();
        return AssignedCondition(sum_with_constant({Elem(&a.v, -one)}, &one));
    }
    AssignedCondition or_(const AssignedCondition& a, const AssignedCondition& b) {  // :428-439
        Fr one = Fr::one();
        Fr c = a.v.val + b.v.val - a.v.val * b.v.val;
        auto cells = one_line_with_last({pr(a.v, one), pr(b.v, one)}, pr(c, -one), nullptr, {-one}, nullptr);
        return AssignedCondition(cells.second);
    }
    AssignedCondition xnor(const AssignedCondition& a, const AssignedCondition& b) {  // :455-467
        Fr one = Fr::one();
        Fr two = one + one;
        Fr c = one - a.v.val - b.v.val + two * a.v.val * b.v.val;
        auto cells = one_line_with_last({pr(a.v, -one), pr(b.v, -one)}, pr(c, -one), &one, {two}, nullptr);
        return AssignedCondition(cells.second);
    }
    // base_chip.rs:574-604 (VAR_COLUMNS >= 5 branch)
    AssignedValue bisec(const AssignedCondition& cond, const AssignedValue& a, const AssignedValue& b) {
        Fr zero = Fr::zero(), one = Fr::one();
        AssignedValue cond_v = cond.v;
        Fr c = cond.v.val * a.val + (one - cond.v.val) * b.val;
        auto cells = one_line_with_last({pr(cond_v, zero), pr(a, zero), pr(cond_v, zero), pr(b, one)}, pr(c, -one),
                                        nullptr, {one, -one}, nullptr);
        return cells.second;
    }
    AssignedCondition bisec_cond(const AssignedCondition& cond, const AssignedCondition& a,
                                 const AssignedCondition& b) {  // :477-485
        return AssignedCondition(bisec(cond, a.v, b.v));
    }
    void assert_true(const AssignedCondition& a) {  // :487-490
        if (!(a.v.val == Fr::one())) throw PanicError("assert_true failed");
        assert_constant(a.v, Fr::one());
    }
    void assert_false(const AssignedCondition& a) {  // :492-495
        if (!(a.v.val == Fr::zero())) throw PanicError("assert_false failed");
        assert_constant(a.v, Fr::zero());
    }
    // base_chip.rs:497-500.  NB: assert_constant's assert_eq! fires before the bool is formed, so
    // in the reference a failing try_assert_false panics rather than returning false.
    bool try_assert_false(const AssignedCondition& a) {
        assert_constant(a.v, Fr::zero());
        return a.v.val == Fr::zero();
    }
};

}  // namespace h2o
