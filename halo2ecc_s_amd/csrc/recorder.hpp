// Host side of the witness engine: a *recording* implementation of the reference's operator surface.
//
// The reference's chips are traits on context structs (`BaseChipOps` src/circuit/base_chip.rs:81-501,
// `RangeChipOps` src/circuit/range_chip.rs:262-348, `SelectChipOps` src/circuit/select_chip.rs:99-162,
// `IntegerChipOps` src/circuit/integer_chip.rs:15-70) that compute values eagerly with BigUint.
// Here the same calls (same names, same argument meaning) record
//   (1) the witness tape the HIP engine replays for the *values* of every advice cell, and
//   (2) everything that is shape-only: fixed cells, assigned/permute flags, the permutation list,
//       heights and offsets (what `Records` holds besides advice values, src/context.rs:241-301).
// Handles carry cell references and the static `times` counter, never values.
//
// Forked contexts (`ParallelClone`, src/circuit/ecc_chip.rs:64-77) are "strands": the body is recorded
// once (strand 0) and replayed by the engine at row offsets strand*delta; for strands >= 1 the body is
// re-run on the host in shape-only mode so fixed cells / permutations are produced exactly as the
// reference's merge() would have collected them.
#pragma once
#include <array>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
#include "hbig.hpp"
#include "tape.h"

namespace h2e {

typedef std::array<uint64_t, 4> FrVal;  // canonical bn256 Fr value, little endian words

struct FrValHash {
    size_t operator()(const FrVal& v) const {
        uint64_t h = 0x9e3779b97f4a7c15ull;
        for (int i = 0; i < 4; i++) h = (h ^ v[i]) * 0xbf58476d1ce4e5b9ull + (h >> 29);
        return (size_t)h;
    }
};

static const char* const BN256_FR_HEX = "30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001";
static const char* const BN256_FQ_HEX = "30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47";
static const char* const BLS12_381_FQ_HEX =
    "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab";
static const char* const BLS12_381_FR_HEX = "73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001";

enum FieldPairId { FP_BN256_FQ_OVER_FR = 0, FP_BLS12_381_FQ_OVER_FR = 1, FP_BLS12_381_FR_OVER_FR = 2 };

static const int LIMB_BITS = 108;        // COMMON_RANGE_BITS * RANGE_VALUE_DECOMPOSE  (range_chip.rs:22-33)
static const int COMMON_BITS = 18;
static const int OVERFLOW_LIMIT = 64;    // 1 << OVERFLOW_BITS                          (context.rs:38)
static const int REDUCE_THRESHOLD = 16;  // 1 << (overflow_bits - 2)                    (integer_chip.rs:376)

// Derived constants of a (W over N) pair: RangeInfo::new (range_info.rs:77-184).
struct FieldPair {
    int id;
    HBig w, n;
    int limbs, w_ceil_bits, d_bits, n_floor_bits;
    int w_lead_bits, d_lead_bits;
    int mul_check_limbs, reduce_check_limbs, pure_w_check_limbs;
    int w_words;
    std::vector<HBig> w_limbs;
    std::vector<std::vector<HBig>> ceil_limbs;  // [times][limb]
    H2EFieldConsts fc;

    FrVal fr(const HBig& x) const {
        HBig r = x % n;
        FrVal v;
        r.to_words(v.data(), 4);
        return v;
    }
    FrVal fr_neg(const HBig& x) const {
        HBig r = x % n;
        if (!r.is_zero()) r = n - r;
        FrVal v;
        r.to_words(v.data(), 4);
        return v;
    }

    explicit FieldPair(int id_) : id(id_) {
        n = HBig::from_hex(BN256_FR_HEX);
        w = HBig::from_hex(id == 0 ? BN256_FQ_HEX : id == 1 ? BLS12_381_FQ_HEX : BLS12_381_FR_HEX);
        w_words = id == 1 ? 6 : 4;
        w_ceil_bits = (w - HBig(1)).bits();
        n_floor_bits = (n - HBig(1)).bits() - 1;
        d_bits = w_ceil_bits + 6 * 2 + 1;                              // range_info.rs:299-314
        limbs = (w_ceil_bits + LIMB_BITS - 1) / LIMB_BITS;
        auto lead = [](int bits) { return bits % LIMB_BITS == 0 ? LIMB_BITS : bits % LIMB_BITS; };
        w_lead_bits = lead(w_ceil_bits);
        d_lead_bits = lead(d_bits);
        // the engine places leading limbs in 2-line range values (context.rs:974-997)
        if (w_lead_bits < 36 || w_lead_bits > 72 || d_lead_bits < 36 || d_lead_bits > 72)
            throw std::runtime_error("leading limb does not fit a 2-line range value");
        pure_w_check_limbs = (w_ceil_bits - n_floor_bits + LIMB_BITS - 1) / LIMB_BITS;
        mul_check_limbs = (std::max(w_ceil_bits * 2 + 12, d_bits + w_ceil_bits) - n_floor_bits + LIMB_BITS - 1) / LIMB_BITS;
        reduce_check_limbs = (std::max(w_ceil_bits + 6, COMMON_BITS + w_ceil_bits) - n_floor_bits + LIMB_BITS - 1) / LIMB_BITS;
        for (int i = 0; i < limbs; i++) w_limbs.push_back(w.shr(i * LIMB_BITS).low_bits(LIMB_BITS));
        // find_w_modulus_of_ceil_times (range_info.rs:334-359)
        ceil_limbs.resize(OVERFLOW_LIMIT);
        HBig w_ceil = HBig(1).shl(w_ceil_bits), limb_modulus = HBig(1).shl(LIMB_BITS);
        for (int t = 1; t < OVERFLOW_LIMIT; t++) {
            HBig max = w_ceil * HBig(t), q, r;
            HBig::divmod(max, w, q, r);
            if (!r.is_zero()) q = q + HBig(1);
            HBig upper = w * q;
            for (int i = 0; i + 1 < limbs; i++) {
                HBig rem = upper.low_bits(LIMB_BITS) + limb_modulus * HBig(t);
                upper = (upper - rem).shr(LIMB_BITS);
                ceil_limbs[t].push_back(rem);
            }
            ceil_limbs[t].push_back(upper);
        }
        // engine constants
        std::memset(&fc, 0, sizeof(fc));
        fc.limbs = limbs;
        fc.w_words = w_words;
        fc.w_bits = w.bits();
        fc.w_ceil_bits = w_ceil_bits;
        fc.d_bits = d_bits;
        fc.w_lead_bits = w_lead_bits;
        fc.d_lead_bits = d_lead_bits;
        fc.mul_check_limbs = mul_check_limbs;
        fc.reduce_check_limbs = reduce_check_limbs;
        fc.pure_w_check_limbs = pure_w_check_limbs;
        fc.barrett_s = 2 * w_ceil_bits + 12;
        fc.input_bytes = w_words * 8;
        w.to_words(fc.w, H2E_W_WORDS_MAX);
        (HBig(1).shl(fc.barrett_s) / w).to_words(fc.w_mu, 8);
        for (int i = 0; i < limbs; i++) w_limbs[i].to_words(fc.w_limbs[i], 2);
        n.to_words(fc.n, 4);
        (HBig(1).shl(512) / n).to_words(fc.n_mu, 5);
        (w % n).to_words(fc.w_native, 4);
        for (int t = 1; t < OVERFLOW_LIMIT; t++) {
            HBig comp;
            for (int i = 0; i < limbs; i++) {
                ceil_limbs[t][i].to_words(fc.ceil_limbs[t][i], 2);
                comp = comp + ceil_limbs[t][i].shl(i * LIMB_BITS);
            }
            (comp % n).to_words(fc.ceil_native[t], 4);
        }
        // Montgomery constants, R = 2^(64 * words)
        auto minv = [](uint64_t p0) {
            uint64_t inv = 1;
            for (int i = 0; i < 63; i++) {
                inv = inv * inv;
                inv = inv * p0;
            }
            return (uint64_t)0 - inv;
        };
        HBig Rw = HBig(1).shl(64 * w_words), Rn = HBig(1).shl(256);
        fc.w_minv = minv(w.word(0));
        ((Rw * Rw) % w).to_words(fc.w_r2, H2E_W_WORDS_MAX);
        (Rw % w).to_words(fc.w_r1, H2E_W_WORDS_MAX);
        fc.n_minv = minv(n.word(0));
        ((Rn * Rn) % n).to_words(fc.n_r2, 4);
        (Rn % n).to_words(fc.n_r1, 4);
        {   // lin_bias (tape.h): digits 2^44 + delta_j of a multiple of w
            HBig ones;
            for (int j = 0; j < 2 * w_words; j++) ones = ones + HBig(1).shl(32 * j);
            HBig base = ones.shl(44);
            HBig delta = (w - (base % w)) % w;
            uint64_t dw[H2E_W_WORDS_MAX];
            delta.to_words(dw, H2E_W_WORDS_MAX);
            for (int j = 0; j < 2 * w_words; j++)
                fc.lin_bias[j] = ((uint64_t)1 << 44) + (uint32_t)(dw[j / 2] >> (32 * (j % 2)));
        }
    }
};

inline const FieldPair& field_pair_of(int id) {
    static std::mutex mu;
    static std::unique_ptr<FieldPair> fps[3];
    if (id < 0 || id > 2) throw std::runtime_error("bad field pair");
    std::lock_guard<std::mutex> g(mu);
    if (!fps[id]) fps[id].reset(new FieldPair(id));
    return *fps[id];
}

// ---- handles (src/assign.rs) --------------------------------------------------------------------
struct AssignedValue {   // assign.rs:25-29 (cell only; the value lives on the device)
    uint32_t ref = H2E_NO_REF;
};
struct AssignedCondition {  // assign.rs:84-85
    AssignedValue v;
};
struct AssignedInteger {  // assign.rs:31-37
    uint32_t limbs_le[H2E_MAX_L] = {H2E_NO_REF, H2E_NO_REF, H2E_NO_REF, H2E_NO_REF};
    uint32_t native = H2E_NO_REF;
    uint64_t times = 1;
    // value tag: hint slot that holds the canonical W value this integer is congruent to (H2E_NO_REF = unknown).
    // A later reduce() of a tagged integer takes its result from that slot in the values-only replay.
    uint32_t vtag = H2E_NO_REF;
    bool vtag_strided = false;
};

enum Chip { BaseChip = 0, RangeChip = 1, SelectChip = 2 };

struct Segment {  // one engine launch
    uint32_t tape_begin = 0, tape_end = 0;
    uint32_t n_strands = 1;
    uint32_t base0 = 0, range0 = 0, select0 = 0;
    uint32_t dbase = 0, drange = 0, dselect = 0;
    uint32_t input_stride = 0;
    uint32_t n_params = 0;
    uint32_t params_begin = 0;
    uint64_t cells = 0;  // advice cells written by this launch, per instance (needs emit_shape)
    uint32_t hint_stride = 0;
    uint32_t fixups_begin = 0, n_fixups = 0;
    uint32_t cuts_begin = 0, n_cuts = 0;  // op indices (relative to tape_begin) where the expansion may be split
    bool is_fork = false;
    int field_pair = 0;   // the W field the segment's integer ops work in (GeneralScalarEccContext has two, context.rs:215-239)
    // scheduling hint: the next segment's value chain is tiny (a handful of waves that need most of a CU's LDS); this
    // segment's bandwidth-bound expansion would starve it, so the expansion is launched behind that chain
    bool expand_after_next = false;
    uint32_t sel_stride = 0;   // selection-buffer entries per strand (segments with pre-selected points)
    bool field_hints = false;  // recorded with Recorder::begin_field_hints: its mul-like ops carry hints of a field-domain predictor
};

struct PreKernel {  // a value-predictor launch that must run before segment `before_segment`
    H2EPreKernel k = {};
    uint32_t before_segment;
    // >= 0: it only depends on the predictors of that (earlier) segment, so it may start right after them on a
    // side stream and overlap that segment's replay (the MSM tail predictor only needs the windows' Jacobian sums)
    int32_t early_after_segment = -1;
};

struct FixedPatch {  // a fixed cell whose value is an instance input (constants made from inputs)
    uint32_t row;        // base row
    uint32_t col;        // fixed column (8 = constant)
    uint32_t input_slot; // W value in the instance's input vector
    int32_t limb;        // limb index, or -1 for "value mod n"
};

struct Offset {  // ecc_chip.rs:36-41
    size_t range_offset_diff = 0, base_offset_diff = 0, select_offset_diff = 0;
    bool operator==(const Offset& o) const {
        return range_offset_diff == o.range_offset_diff && base_offset_diff == o.base_offset_diff &&
               select_offset_diff == o.select_offset_diff;
    }
};

// Context + Records (shape part) + BaseChipOps/RangeChipOps/SelectChipOps/IntegerChipOps, recording.
struct Recorder {
    FieldPair fp;          // the integer chip's current W field (a copy: use_field switches it)
    int primary_field;
    // ---- program ----
    std::vector<H2EOp> tape;
    std::vector<uint32_t> aux;
    std::vector<uint64_t> pool;
    std::vector<uint32_t> params;
    std::vector<Segment> segments;
    std::vector<FixedPatch> fixed_patches;
    std::vector<uint32_t> outputs;  // absolute refs of the workload's result cells (program specific)
    std::vector<uint32_t> fixups;   // per segment: strand-relative base rows of the second invert row (see tape.h)
    std::vector<uint32_t> cuts;     // per segment: op indices where the full expansion may be split into sub-ranges
    uint32_t cur_tape_begin = 0;
    std::vector<PreKernel> pre_kernels;
    std::vector<uint32_t> pre_args;
    uint32_t n_hint_slots = 0, n_jac_slots = 0, n_sel_slots = 0;
    // hinted-division state: while hint_on, every int_div takes its quotient from slot hint_base + hint_count++
    bool hint_on = false;
    uint32_t hint_base = 0, hint_count = 0;
    // hint_mode 2 ("full value hints", MSM chains): every ecc op owns a block of H2E_ECC_HINT_SLOTS slots that the
    // predictor + finalize kernels fill with the canonical value of every mul-like result of that op (tape.h);
    // hint_count then counts blocks and block `n_blocks` of a strand holds the chain's initial point.
    // hint_mode 3 ("field hints", the pairings): every int_mul / reduce / int_div of the main context gets a hint slot of
    // its own; a field-domain predictor (h2e_capi.cpp compile_field_chain, engine.hip h2e_field_chain) walks the same
    // computation as plain arithmetic mod w - where a reduce is free and a chain of additions is one linear combination -
    // and fills the slots with the canonical values.
    int hint_mode = 0;
    uint32_t ecc_block = H2E_NO_REF;   // first slot of the current ecc op's block
    uint32_t next_vtag = H2E_NO_REF;   // tag for the result of the next integer op
    uint32_t n_input_slots = 0;
    uint64_t seg_cells_start = 0;
    // ---- shape artefacts (Records minus advice values) ----
    bool emit_shape = true;
    std::vector<FrVal> dict;  // dict[0] unused: index 0 = None
    std::unordered_map<FrVal, uint32_t, FrValHash> dict_map;
    std::vector<uint32_t> base_fix;    // [row][9] dictionary ids
    std::vector<uint32_t> range_fix;   // [row][2]
    std::vector<uint32_t> select_fix;  // [row][2]
    std::vector<uint8_t> base_flags;   // [row][5] bit0 assigned, bit1 permute
    std::vector<uint8_t> range_flags;  // [row][3]
    std::vector<uint8_t> select_flags; // [row][2]
    std::vector<std::pair<uint32_t, uint32_t>> permutations;  // absolute cell refs
    size_t base_height = 0, range_height = 0, select_height = 0;
    size_t n_advice_cells = 0;
    // ---- cursors (Context, context.rs:40-46) ----
    size_t base_offset = 0, range_offset = 0, select_offset = 0;
    // ---- strand state ----
    bool in_strand = false, record_tape = true;
    size_t strand_off[3] = {0, 0, 0};
    uint32_t strand_index = 0, strand_param_cursor = 0, strand_n_params = 0, strand_params_begin = 0;
    // ---- dictionary ids of recurring fixed values ----
    uint32_t id_zero, id_one, id_neg_one, id_two, id_four, id_three;
    uint32_t id_limb_coeff[H2E_MAX_L], id_w_native, id_neg_w_native, id_neg_w_limb[H2E_MAX_L], id_w_limb[H2E_MAX_L];
    uint32_t id_ceil[OVERFLOW_LIMIT][H2E_MAX_L];
    uint32_t id_neg_limb_modulus, id_K0, id_K1, id_reduce_k[2], id_small[16], id_pow2[8], id_tag[19];

    explicit Recorder(const FieldPair& f) : fp(f), primary_field(f.id) {
        dict.push_back(FrVal{0, 0, 0, 0});
        id_zero = intern(fp.fr(HBig(0)));
        id_one = intern(fp.fr(HBig(1)));
        id_neg_one = intern(fp.fr_neg(HBig(1)));
        id_two = intern(fp.fr(HBig(2)));
        id_three = intern(fp.fr(HBig(3)));
        id_four = intern(fp.fr(HBig(4)));
        HBig limb_modulus = HBig(1).shl(LIMB_BITS);
        for (int i = 0; i < H2E_MAX_L; i++) id_limb_coeff[i] = intern(fp.fr(HBig(1).shl(i * LIMB_BITS)));
        id_neg_limb_modulus = intern(fp.fr_neg(limb_modulus));
        id_reduce_k[0] = intern(fp.fr(limb_modulus * HBig(OVERFLOW_LIMIT)));                         // :362-365
        id_reduce_k[1] = intern(fp.fr(limb_modulus * HBig(OVERFLOW_LIMIT) - HBig(OVERFLOW_LIMIT)));
        for (int k = 0; k < 16; k++) id_small[k] = intern(fp.fr(HBig((uint64_t)k)));
        for (int k = 0; k < 8; k++) id_pow2[k] = intern(fp.fr(HBig(1ull << k)));
        for (int k = 0; k <= 18; k++) id_tag[k] = intern(fp.fr(HBig((uint64_t)k)));
        field_ids();
        begin_segment();
    }

    // dictionary ids of the fixed values that depend on W
    void field_ids() {
        HBig limb_modulus = HBig(1).shl(LIMB_BITS);
        for (int i = 0; i < fp.limbs; i++) {
            id_w_limb[i] = intern(fp.fr(fp.w_limbs[i]));
            id_neg_w_limb[i] = intern(fp.fr_neg(fp.w_limbs[i]));
        }
        id_w_native = intern(fp.fr(fp.w));
        id_neg_w_native = intern(fp.fr_neg(fp.w));
        for (int t = 1; t < OVERFLOW_LIMIT; t++)
            for (int i = 0; i < fp.limbs; i++) id_ceil[t][i] = intern(fp.fr(fp.ceil_limbs[t][i]));
        HBig borrow = HBig((uint64_t)fp.limbs) * limb_modulus + HBig(2);     // integer_chip.rs:112
        id_K0 = intern(fp.fr(limb_modulus * borrow));                       // :117
        id_K1 = intern(fp.fr(limb_modulus * borrow - borrow));              // :144
    }
    // Switch the integer chip to another W field (the scalar_integer_ctx / base_integer_ctx of a
    // GeneralScalarEccContext share one Context, context.rs:215-239).  Main context only: a segment (= one engine
    // launch = one kernel instantiation) works in one field.
    void use_field(int id) {
        if (id == fp.id) return;
        if (in_strand) throw std::runtime_error("use_field inside a fork");
        fp = field_pair_of(id);
        field_ids();
        close_segment();
        begin_segment();
    }

    uint32_t intern(const FrVal& v) {
        auto it = dict_map.find(v);
        if (it != dict_map.end()) return it->second;
        uint32_t id = (uint32_t)dict.size();
        dict.push_back(v);
        dict_map.emplace(v, id);
        return id;
    }

    // ---- segments / strands --------------------------------------------------------------------
    void begin_segment() {
        Segment s;
        s.field_pair = fp.id;
        s.tape_begin = s.tape_end = (uint32_t)tape.size();
        s.fixups_begin = (uint32_t)fixups.size();
        s.cuts_begin = (uint32_t)cuts.size();
        cur_tape_begin = s.tape_begin;
        segments.push_back(s);
        seg_cells_start = n_advice_cells;
    }
    void close_segment() {
        segments.back().tape_end = (uint32_t)tape.size();
        segments.back().cells = n_advice_cells - seg_cells_start;
        segments.back().n_fixups = (uint32_t)fixups.size() - segments.back().fixups_begin;
        segments.back().n_cuts = (uint32_t)cuts.size() - segments.back().cuts_begin;
    }
    // End the main context's current segment here and open the next one: what follows becomes a launch of its own, with its own
    // kind of value chain (the MSM's accumulation loop - every mul-like result hinted - ahead of the unhinted rest of the call).
    void split_segment() {
        if (in_strand) throw std::runtime_error("split_segment inside a fork");
        if (tape.size() == cur_tape_begin) return;
        close_segment();
        begin_segment();
    }
    // Mark a point where the full expansion of the current segment may be split (see H2ELaunch.sub).
    void cut() {
        if (!record_tape) return;
        uint32_t at = (uint32_t)tape.size() - cur_tape_begin;
        if (at == 0) return;
        if (!cuts.empty() && cuts.size() > segments_cut_floor() && cuts.back() == at) return;
        cuts.push_back(at);
    }
    size_t segments_cut_floor() const { return in_strand ? fork_cuts_begin : segments.back().cuts_begin; }
    uint32_t fork_cuts_begin = 0;
    // hinted divisions (quotients predicted by the V kernels): slots are consecutive in call order
    void begin_hints(uint32_t base, int mode = 1) {
        hint_on = true;
        hint_mode = mode;
        hint_base = base;
        hint_count = 0;
    }
    uint32_t end_hints() {
        hint_on = false;
        hint_mode = 0;
        ecc_block = next_vtag = H2E_NO_REF;
        return hint_count;
    }
    uint32_t hint_slots_used() const { return hint_mode == 2 ? H2E_ECC_HINT_SLOTS * (hint_count + 1) : hint_count; }
    bool field_hints() const { return hint_on && hint_mode == 3 && !in_strand; }
    // program level: switch field hints on for everything recorded from here (main context)
    void begin_field_hints() {
        if (hint_on) throw std::runtime_error("begin_field_hints: hints already on");
        begin_hints(n_hint_slots, 3);
    }
    void end_field_hints() {
        if (hint_on && hint_mode == 3) n_hint_slots += end_hints();
    }
    void begin_ecc_op() {
        if (hint_on && hint_mode == 2) ecc_block = hint_base + H2E_ECC_HINT_SLOTS * hint_count++;
    }
    void end_ecc_op() { ecc_block = next_vtag = H2E_NO_REF; }
    void expect(uint32_t k) {  // the next integer result is congruent to slot k of the current ecc block
        if (ecc_block != H2E_NO_REF) next_vtag = ecc_block + k;
    }
    void take_vtag(AssignedInteger& r) {
        r.vtag = next_vtag;
        r.vtag_strided = in_strand;
        next_vtag = H2E_NO_REF;
    }

    Offset offset() const {
        Offset o;
        o.base_offset_diff = base_offset;
        o.range_offset_diff = range_offset;
        o.select_offset_diff = select_offset;
        return o;
    }

    // Run `body(strand)` for n strands as a forked region.  Every strand must consume the same Offset
    // (the reference asserts this for MSM windows, ecc_chip.rs:339).  Returns the per-strand Offset.
    // `merged` = the strands are the reference's cloned contexts that are merge()d back (the MSM windows,
    // ecc_chip.rs:289-352): merge sets range_height = max(select_height, other.range_height) with the parent's select
    // height *before* that clone's select rows are merged (quirk Q3, native_scalar_ecc_chip.rs:80-89)
    Offset fork(uint32_t n_strands, uint32_t input_stride, const std::function<void(uint32_t)>& body, bool merged = false) {
        if (in_strand) throw std::runtime_error("nested fork");
        Offset delta;
        if (n_strands == 0) return delta;
        close_segment();
        Segment seg;
        seg.tape_begin = (uint32_t)tape.size();
        seg.is_fork = true;
        seg.field_pair = fp.id;
        seg.n_strands = n_strands;
        seg.base0 = (uint32_t)base_offset;
        seg.range0 = (uint32_t)range_offset;
        seg.select0 = (uint32_t)select_offset;
        seg.input_stride = input_stride;
        seg.params_begin = (uint32_t)params.size();
        seg.fixups_begin = (uint32_t)fixups.size();
        seg.cuts_begin = fork_cuts_begin = (uint32_t)cuts.size();
        cur_tape_begin = seg.tape_begin;
        size_t b0 = base_offset, r0 = range_offset, s0 = select_offset;
        const size_t select_height_at_fork = select_height;
        seg_cells_start = n_advice_cells;
        in_strand = true;
        strand_params_begin = seg.params_begin;
        for (uint32_t k = 0; k < n_strands; k++) {
            strand_index = k;
            strand_off[0] = base_offset = b0 + k * delta.base_offset_diff;
            strand_off[1] = range_offset = r0 + k * delta.range_offset_diff;
            strand_off[2] = select_offset = s0 + k * delta.select_offset_diff;
            strand_param_cursor = 0;
            record_tape = (k == 0);
            hint_count = 0;
            if (k > 0 && !emit_shape && strand_n_params == 0) break;  // nothing left to learn from further strands
            if (k > 0) params.resize(params.size() + strand_n_params, H2E_NO_REF);
            body(k);
            if (k == 0) {
                delta.base_offset_diff = base_offset - b0;
                delta.range_offset_diff = range_offset - r0;
                delta.select_offset_diff = select_offset - s0;
                strand_n_params = strand_param_cursor;
                seg.hint_stride = hint_on ? hint_slots_used() : 0;
            } else {
                Offset d;
                d.base_offset_diff = base_offset - strand_off[0];
                d.range_offset_diff = range_offset - strand_off[1];
                d.select_offset_diff = select_offset - strand_off[2];
                if (!(d == delta)) throw std::runtime_error("fork: strands consume different offsets");
                if (strand_param_cursor != strand_n_params) throw std::runtime_error("fork: strands use different parameter counts");
            }
        }
        in_strand = false;
        record_tape = true;
        seg.tape_end = (uint32_t)tape.size();
        seg.dbase = (uint32_t)delta.base_offset_diff;
        seg.drange = (uint32_t)delta.range_offset_diff;
        seg.dselect = (uint32_t)delta.select_offset_diff;
        seg.n_params = strand_n_params;
        seg.n_fixups = (uint32_t)fixups.size() - seg.fixups_begin;
        seg.n_cuts = (uint32_t)cuts.size() - seg.cuts_begin;
        seg.cells = n_advice_cells - seg_cells_start;
        segments.push_back(seg);
        // apply_offset_diff(delta.scale(n)) (ecc_chip.rs:352)
        base_offset = b0 + n_strands * delta.base_offset_diff;
        range_offset = r0 + n_strands * delta.range_offset_diff;
        select_offset = s0 + n_strands * delta.select_offset_diff;
        if (!emit_shape) {
            // heights as the sequential run would leave them
            if (delta.base_offset_diff) base_height = std::max(base_height, base_offset);
            if (delta.range_offset_diff) range_height = std::max(range_height, range_offset + 1);
            if (delta.select_offset_diff) select_height = std::max(select_height, select_offset);
        }
        if (merged) {
            // select height of the parent when the last clone is merged = after the clones before it
            size_t sel_before_last = delta.select_offset_diff ? std::max<size_t>(select_height_at_fork, s0 + (n_strands - 1) * delta.select_offset_diff)
                                                              : select_height_at_fork;
            range_height = std::max(range_height, sel_before_last);
        }
        strand_n_params = 0;
        begin_segment();
        return delta;
    }

    // Inside a fork body: turn a reference that differs per strand into a strand parameter.
    uint32_t param(uint32_t abs_ref) {
        if (!in_strand) return abs_ref;
        uint32_t p = strand_param_cursor++;
        if (strand_index == 0) params.push_back(abs_ref);
        else params[strand_params_begin + (size_t)strand_index * strand_n_params + p] = abs_ref;
        return H2E_MAKE_REF(H2E_REGION_PARAM, 0, 0, p);
    }
    AssignedValue param(const AssignedValue& v) { return AssignedValue{param(v.ref)}; }
    AssignedInteger param(const AssignedInteger& a) {
        AssignedInteger r = a;
        for (int i = 0; i < fp.limbs; i++) r.limbs_le[i] = param(a.limbs_le[i]);
        r.native = param(a.native);
        return r;
    }
    // Shift a strand-0 relative handle to the absolute cells of strand k of a finished fork.
    uint32_t strand_ref(uint32_t ref, const Segment& seg, uint32_t k) const {
        if (H2E_REF_REGION(ref) == H2E_REGION_PARAM || !H2E_REF_REL(ref)) return ref;
        uint32_t region = H2E_REF_REGION(ref);
        uint32_t off = region == 0 ? seg.base0 + k * seg.dbase : region == 1 ? seg.range0 + k * seg.drange : seg.select0 + k * seg.dselect;
        return H2E_MAKE_REF(region, H2E_REF_COL(ref), 0, H2E_REF_ROW(ref) + off);
    }
    AssignedInteger strand_int(const AssignedInteger& a, const Segment& seg, uint32_t k) const {
        AssignedInteger r = a;
        for (int i = 0; i < fp.limbs; i++) r.limbs_le[i] = strand_ref(a.limbs_le[i], seg, k);
        r.native = strand_ref(a.native, seg, k);
        r.vtag = H2E_NO_REF;
        return r;
    }

    // ---- cell plumbing -------------------------------------------------------------------------
    uint32_t mk(int region, int col, size_t abs_row) const {
        if (in_strand) return H2E_MAKE_REF(region, col, 1, abs_row - strand_off[region]);
        return H2E_MAKE_REF(region, col, 0, abs_row);
    }
    uint32_t resolve_param_aware(uint32_t ref) const {
        if (H2E_REF_REGION(ref) == H2E_REGION_PARAM) {
            size_t p = H2E_REF_ROW(ref);
            uint32_t r = strand_index == 0 ? params[strand_params_begin + p]
                                           : params[strand_params_begin + (size_t)strand_index * strand_n_params + p];
            return r;
        }
        if (H2E_REF_REL(ref)) {
            uint32_t region = H2E_REF_REGION(ref);
            return H2E_MAKE_REF(region, H2E_REF_COL(ref), 0, H2E_REF_ROW(ref) + strand_off[region]);
        }
        return ref;
    }
    void grow(std::vector<uint32_t>& v, size_t n) {
        if (v.size() < n) v.resize(std::max(n, v.size() * 2), 0);
    }
    void grow8(std::vector<uint8_t>& v, size_t n) {
        if (v.size() < n) v.resize(std::max(n, v.size() * 2), 0);
    }
    uint8_t& flags_of(uint32_t abs_ref) {
        uint32_t region = H2E_REF_REGION(abs_ref), col = H2E_REF_COL(abs_ref);
        size_t row = H2E_REF_ROW(abs_ref);
        if (region == 0) {
            grow8(base_flags, (row + 1) * 5);
            return base_flags[row * 5 + col];
        }
        if (region == 1) {
            grow8(range_flags, (row + 1) * 3);
            return range_flags[row * 3 + col];
        }
        grow8(select_flags, (row + 1) * 2);
        return select_flags[row * 2 + col];
    }
    void set_assigned(uint32_t abs_ref) {
        uint8_t& f = flags_of(abs_ref);
        if (!(f & 1)) n_advice_cells++;
        f |= 1;
    }
    void permute(uint32_t src_ref, uint32_t new_abs) {  // context.rs:648-656
        uint32_t src_abs = resolve_param_aware(src_ref);
        flags_of(new_abs) |= 2;
        flags_of(src_abs) |= 2;
        permutations.push_back(std::make_pair(src_abs, new_abs));
    }

    // ---- L0 row writers (shape part of context.rs:634-997) ---------------------------------------
    struct Col {
        uint32_t src;    // source cell ref, or H2E_NO_REF for a bare value
        uint32_t coeff;  // dictionary id of the fixed coefficient
        bool used;
    };
    static Col A(uint32_t ref, uint32_t coeff) { return Col{ref, coeff, true}; }  // pair!(&assigned, coeff)
    static Col U(uint32_t coeff) { return Col{H2E_NO_REF, coeff, true}; }        // pair!(value, coeff)
    static Col none() { return Col{H2E_NO_REF, 0, false}; }

    // one_line / one_line_with_last (context.rs:634-714).  cols[0..3] left-to-right pairs, `last` = col 4.
    // mul0/mul1/next/constant: dictionary ids, 0 = None.  Returns the base row written.
    size_t base_line(const Col* cols, int ncols, const Col& last, uint32_t mul0, uint32_t mul1, uint32_t next,
                     uint32_t constant) {
        size_t row = base_offset;
        base_offset += 1;
        if (!emit_shape) {
            if (row >= base_height) base_height = row + 1;
            return row;
        }
        if (row >= base_height) base_height = row + 1;
        grow(base_fix, (row + 1) * 9);
        for (int i = 0; i < ncols; i++) {
            uint32_t cell = H2E_MAKE_REF(0, i, 0, row);
            if (cols[i].src != H2E_NO_REF) permute(cols[i].src, cell);
            set_assigned(cell);
            base_fix[row * 9 + i] = cols[i].coeff;
        }
        if (mul0) base_fix[row * 9 + 5] = mul0;
        if (mul1) base_fix[row * 9 + 6] = mul1;
        if (next) base_fix[row * 9 + 7] = next;
        if (constant) base_fix[row * 9 + 8] = constant;
        if (last.used) {
            uint32_t cell = H2E_MAKE_REF(0, 4, 0, row);
            if (last.src != H2E_NO_REF) permute(last.src, cell);
            set_assigned(cell);
            base_fix[row * 9 + 4] = last.coeff;
        }
        return row;
    }
    size_t base_line(std::initializer_list<Col> cols, const Col& last, uint32_t mul0 = 0, uint32_t mul1 = 0,
                     uint32_t next = 0, uint32_t constant = 0) {
        return base_line(cols.begin(), (int)cols.size(), last, mul0, mul1, next, constant);
    }

    void range_cell(size_t row, int col) {
        set_assigned(H2E_MAKE_REF(1, col, 0, row));
    }
    void range_fix_set(size_t row, int col, uint32_t id) {
        grow(range_fix, (row + 1) * 2);
        range_fix[row * 2 + col] = id;
    }
    // assign_{one,two,three}_line_range_value (context.rs:835-972); returns the acc cell row
    size_t range_value(int lines, int bits) {
        size_t row = range_offset;
        range_offset += lines;
        if (row + lines >= range_height) range_height = row + lines + 1;  // ensure_range_record_size (quirk Q4)
        if (!emit_shape) return row;
        if (lines == 1) {
            range_fix_set(row, 0, id_one);
            range_fix_set(row, 1, id_tag[bits]);
            range_cell(row, 1);
            range_cell(row, 0);
        } else if (lines == 2) {
            range_fix_set(row, 0, id_two);
            range_cell(row, 2);
            range_cell(row + 1, 2);
            int t0 = bits >= 3 * COMMON_BITS ? COMMON_BITS : bits % COMMON_BITS;
            range_fix_set(row, 1, id_tag[t0]);
            range_cell(row, 1);
            int t1 = bits > 3 * COMMON_BITS ? bits - 3 * COMMON_BITS : 0;
            range_fix_set(row + 1, 1, id_tag[t1]);
            range_cell(row + 1, 1);
            range_cell(row, 0);
        } else {
            range_fix_set(row, 0, id_three);
            range_cell(row, 2);
            range_cell(row + 1, 2);
            range_cell(row + 2, 2);
            int t0 = bits >= 4 * COMMON_BITS ? COMMON_BITS : bits % COMMON_BITS;
            range_fix_set(row, 1, id_tag[t0]);
            range_cell(row, 1);
            int t1 = bits >= 5 * COMMON_BITS ? COMMON_BITS : bits > 4 * COMMON_BITS ? bits % COMMON_BITS : 0;
            range_fix_set(row + 1, 1, id_tag[t1]);
            range_cell(row + 1, 1);
            int t2 = bits > 5 * COMMON_BITS ? bits - 5 * COMMON_BITS : 0;
            range_fix_set(row + 2, 1, id_tag[t2]);
            range_cell(row + 2, 1);
            range_cell(row, 0);
        }
        return row;
    }
    // RangeChipOps (range_chip.rs:287-347): return the acc cell
    uint32_t assign_common() { return mk(1, 0, range_value(1, COMMON_BITS)); }
    uint32_t assign_nonleading_limb() { return mk(1, 0, range_value(3, LIMB_BITS)); }
    uint32_t assign_w_ceil_leading_limb() { return mk(1, 0, range_value(2, fp.w_ceil_bits % LIMB_BITS)); }
    uint32_t assign_d_leading_limb() { return mk(1, 0, range_value(2, fp.d_bits % LIMB_BITS)); }

    // SelectChipOps (select_chip.rs:124-161, context.rs:749-801)
    FrVal encode_offset(size_t g, size_t offset, size_t limb_offset) const {  // select_chip.rs:118-122
        return FrVal{(uint64_t)limb_offset, (uint64_t)g, (uint64_t)offset, 0};
    }
    void assign_cache_value(uint32_t v_ref, size_t offset, size_t group_index, size_t selector) {
        size_t row = select_offset;
        select_offset += 1;
        if (row >= select_height) select_height = row + 1;
        if (!emit_shape) return;
        uint32_t cell = H2E_MAKE_REF(2, 0, 0, row);
        set_assigned(cell);
        // permutations.push((idx, v.cell)) — select cell first (context.rs:760)
        uint32_t src_abs = resolve_param_aware(v_ref);
        permutations.push_back(std::make_pair(cell, src_abs));
        flags_of(cell) |= 2;
        flags_of(src_abs) |= 2;
        grow(select_fix, (row + 1) * 2);
        select_fix[row * 2 + 0] = intern(encode_offset(group_index, selector, offset));
        select_fix[row * 2 + 1] = id_zero;
    }
    uint32_t assign_selected_value(size_t offset, size_t group_index, uint32_t selector_ref) {
        size_t row = select_offset;
        select_offset += 1;
        if (row >= select_height) select_height = row + 1;
        if (emit_shape) {
            set_assigned(H2E_MAKE_REF(2, 0, 0, row));
            uint32_t sel_cell = H2E_MAKE_REF(2, 1, 0, row);
            set_assigned(sel_cell);
            uint32_t src_abs = resolve_param_aware(selector_ref);
            permutations.push_back(std::make_pair(sel_cell, src_abs));
            flags_of(sel_cell) |= 2;
            flags_of(src_abs) |= 2;
            grow(select_fix, (row + 1) * 2);
            select_fix[row * 2 + 0] = intern(encode_offset(group_index, 0, offset));
            select_fix[row * 2 + 1] = id_one;
        }
        return mk(2, 0, row);
    }

    // ---- tape helpers --------------------------------------------------------------------------
    H2EOp new_op(uint16_t opcode, uint32_t imm = 0, uint16_t flags = 0) const {
        H2EOp op;
        std::memset(&op, 0, sizeof(op));
        op.opcode = opcode;
        op.flags = flags;
        op.imm = imm;
        op.base_row = (uint32_t)(base_offset - (in_strand ? strand_off[0] : 0));
        op.range_row = (uint32_t)(range_offset - (in_strand ? strand_off[1] : 0));
        op.select_row = (uint32_t)(select_offset - (in_strand ? strand_off[2] : 0));
        for (int i = 0; i < H2E_OP_MAX_REFS; i++) op.refs[i] = H2E_NO_REF;
        return op;
    }
    // auto_cut_every > 0: in the main context, allow the expansion to be split every that many ops (any op
    // boundary is a valid cut once the values-only replay has run; see H2ELaunch.sub)
    uint32_t auto_cut_every = 0;
    void push(const H2EOp& op) {
        if (!record_tape) return;
        tape.push_back(op);
        if (hint_on && hint_mode == 3 && !in_strand) segments.back().field_hints = true;
        if (auto_cut_every && !in_strand) {
            uint32_t at = (uint32_t)tape.size() - cur_tape_begin;
            uint32_t last = cuts.size() > segments.back().cuts_begin ? cuts.back() : 0;
            // (never between a SUM_LIMBS and the ASSERT_CONST that reads its cell: no value chain stores that cell - it is an
            // expansion-only row pair of assert_int_equal, integer_chip.rs:607-611 - so the two stay in one sub-range)
            if (at - last >= auto_cut_every && op.opcode != H2E_OP_SUM_LIMBS) cuts.push_back(at);
        }
    }
    void put_int(H2EOp& op, int at, const AssignedInteger& a) const {
        for (int i = 0; i < fp.limbs; i++) op.refs[at + i] = a.limbs_le[i];
        op.refs[at + fp.limbs] = a.native;
    }
    uint32_t pool_fr(const FrVal& v) {
        uint32_t at = (uint32_t)pool.size();
        if (record_tape)
            for (int i = 0; i < 4; i++) pool.push_back(v[i]);
        return at;
    }
    uint32_t pool_w(const HBig& x) {
        uint32_t at = (uint32_t)pool.size();
        if (record_tape)
            for (int i = 0; i < fp.w_words; i++) pool.push_back(x.word(i));
        return at;
    }
    uint32_t alloc_inputs(uint32_t n) {
        uint32_t at = n_input_slots;
        n_input_slots += n;
        return at;
    }

    // =============================================================================================
    // BaseChipOps (base_chip.rs:81-605)
    // =============================================================================================
    // assign_constant (base_chip.rs:344-349): [v * -1], constant v
    AssignedValue assign_constant(const FrVal& v) {
        H2EOp op = new_op(H2E_OP_CONST, pool_fr(v));
        push(op);
        size_t row = base_line({U(id_neg_one)}, none(), 0, 0, 0, emit_shape ? intern(v) : 1);
        return AssignedValue{mk(0, 0, row)};
    }
    AssignedValue assign_constant_u64(uint64_t v) { return assign_constant(FrVal{v, 0, 0, 0}); }
    // assign (base_chip.rs:351-355): value from input slot
    AssignedValue assign(uint32_t input_slot, bool strided = false) {
        push(new_op(H2E_OP_ASSIGN, input_slot, strided ? H2E_FLAG_INPUT_STRIDED : 0));
        size_t row = base_line({U(id_zero)}, none());
        return AssignedValue{mk(0, 0, row)};
    }
    // assign_bit (base_chip.rs:357-367): [a*1, a*0], mul0 = -1   (quirk Q2: two unconstrained copies)
    AssignedCondition assign_bit(uint32_t input_slot, bool strided = false) {
        push(new_op(H2E_OP_ASSIGN_BIT, input_slot, strided ? H2E_FLAG_INPUT_STRIDED : 0));
        size_t row = base_line({U(id_one), U(id_zero)}, none(), id_neg_one);
        return AssignedCondition{AssignedValue{mk(0, 0, row)}};
    }
    // assert_constant (base_chip.rs:375-379) for b in {0, 1}
    void assert_constant(const AssignedValue& a, uint64_t b, uint16_t flags = 0) {
        H2EOp op = new_op(H2E_OP_ASSERT_CONST, (uint32_t)b, flags);
        op.refs[0] = a.ref;
        push(op);
        base_line({A(a.ref, id_neg_one)}, none(), 0, 0, 0, b ? id_one : id_zero);
    }
    void assert_true(const AssignedCondition& a) { assert_constant(a.v, 1); }    // :487-490
    void assert_false(const AssignedCondition& a) { assert_constant(a.v, 0); }   // :492-495
    // try_assert_false (:497-500): a failing instance is reported through the status word
    void try_assert_false(const AssignedCondition& a, uint16_t unsafe_flag) { assert_constant(a.v, 0, unsafe_flag); }
    AssignedCondition and_(const AssignedCondition& a, const AssignedCondition& b) {  // :392-396 -> mul :176-193
        H2EOp op = new_op(H2E_OP_AND);
        op.refs[0] = a.v.ref;
        op.refs[1] = b.v.ref;
        push(op);
        size_t row = base_line({A(a.v.ref, id_zero), A(b.v.ref, id_zero)}, U(id_neg_one), id_one);
        return AssignedCondition{AssignedValue{mk(0, 4, row)}};
    }
    AssignedCondition not_(const AssignedCondition& a) {  // :398-403
        H2EOp op = new_op(H2E_OP_NOT);
        op.refs[0] = a.v.ref;
        push(op);
        size_t row = base_line({A(a.v.ref, id_neg_one)}, U(id_neg_one), 0, 0, 0, id_one);
        return AssignedCondition{AssignedValue{mk(0, 4, row)}};
    }
    AssignedCondition or_(const AssignedCondition& a, const AssignedCondition& b) {  // :428-439
        H2EOp op = new_op(H2E_OP_OR);
        op.refs[0] = a.v.ref;
        op.refs[1] = b.v.ref;
        push(op);
        size_t row = base_line({A(a.v.ref, id_one), A(b.v.ref, id_one)}, U(id_neg_one), id_neg_one);
        return AssignedCondition{AssignedValue{mk(0, 4, row)}};
    }
    AssignedCondition xnor(const AssignedCondition& a, const AssignedCondition& b) {  // :455-467
        H2EOp op = new_op(H2E_OP_XNOR);
        op.refs[0] = a.v.ref;
        op.refs[1] = b.v.ref;
        push(op);
        size_t row = base_line({A(a.v.ref, id_neg_one), A(b.v.ref, id_neg_one)}, U(id_neg_one), id_two, 0, 0, id_one);
        return AssignedCondition{AssignedValue{mk(0, 4, row)}};
    }
    // bisec (base_chip.rs:574-604): [cond*0, a*0, cond*0, b*1 | c*-1], mul = (1, -1)
    size_t bisec_row(uint32_t cond, uint32_t a, uint32_t b) {
        return base_line({A(cond, id_zero), A(a, id_zero), A(cond, id_zero), A(b, id_one)}, U(id_neg_one), id_one, id_neg_one);
    }
    AssignedValue bisec(const AssignedCondition& cond, const AssignedValue& a, const AssignedValue& b) {
        H2EOp op = new_op(H2E_OP_BISEC);
        op.refs[0] = cond.v.ref;
        op.refs[1] = a.ref;
        op.refs[2] = b.ref;
        push(op);
        return AssignedValue{mk(0, 4, bisec_row(cond.v.ref, a.ref, b.ref))};
    }
    AssignedCondition bisec_cond(const AssignedCondition& cond, const AssignedCondition& a, const AssignedCondition& b) {
        return AssignedCondition{bisec(cond, a.v, b.v)};  // :477-485
    }

    // =============================================================================================
    // IntegerChipOps (integer_chip.rs:227-686)
    // =============================================================================================
    // shape of assign_w / assign_d (integer_chip.rs:236-281); returns handle of the new integer
    AssignedInteger shape_assigned(bool is_d) {
        AssignedInteger r;
        for (int i = 0; i + 1 < fp.limbs; i++) r.limbs_le[i] = assign_nonleading_limb();
        r.limbs_le[fp.limbs - 1] = is_d ? assign_d_leading_limb() : assign_w_ceil_leading_limb();
        Col cols[H2E_MAX_L];
        for (int i = 0; i < fp.limbs; i++) cols[i] = A(r.limbs_le[i], id_limb_coeff[i]);
        size_t row = base_line(cols, fp.limbs, U(id_neg_one), 0, 0, 0, 0);
        r.native = mk(0, 4, row);
        r.times = 1;
        return r;
    }
    // native = sum_with_constant(limbs (.) limb_coeffs) row shared by add/sub/neg/mul_small (e.g. :397-401)
    uint32_t shape_native_row(const uint32_t* limb_cells) {
        Col cols[H2E_MAX_L];
        for (int i = 0; i < fp.limbs; i++) cols[i] = A(limb_cells[i], id_limb_coeff[i]);
        size_t row = base_line(cols, fp.limbs, U(id_neg_one), 0, 0, 0, 0);
        return mk(0, 4, row);
    }
    // mul-equation rows (integer_chip.rs:73-215); d = limb cells of the quotient
    void shape_mul_equation(const AssignedInteger& a, const AssignedInteger& b, const AssignedInteger& d,
                            const AssignedInteger& rem) {
        if (!(a.times < (uint64_t)OVERFLOW_LIMIT) || !(b.times < (uint64_t)OVERFLOW_LIMIT) || rem.times != 1)
            throw std::runtime_error("mul equation: times out of range (integer_chip.rs:80-82)");
        int L = fp.limbs;
        std::vector<uint32_t> lcells;
        for (int pos = 0; pos < fp.mul_check_limbs; pos++) {
            int r_bound = std::min(pos + 1, L), l_bound = pos >= L - 1 ? pos - (L - 1) : 0;
            if (r_bound - l_bound == 1) {
                int i = l_bound;  // mul_add (base_chip.rs:219-243)
                size_t row = base_line({A(a.limbs_le[i], id_zero), A(b.limbs_le[pos - i], id_zero), A(d.limbs_le[i], id_neg_w_limb[pos - i])},
                                       U(id_neg_one), id_one);
                lcells.push_back(mk(0, 4, row));
            } else {  // mul_add_with_next_line (base_chip.rs:245-281)
                for (int i = l_bound; i < r_bound; i++)
                    base_line({A(a.limbs_le[i], id_zero), A(b.limbs_le[pos - i], id_zero), A(d.limbs_le[i], id_neg_w_limb[pos - i])},
                              U(i == l_bound ? id_zero : id_one), id_one, 0, id_neg_one);
                size_t row = base_line({}, U(id_zero));
                lcells.push_back(mk(0, 4, row));
            }
        }
        uint32_t v_h = H2E_NO_REF, v_l = H2E_NO_REF;
        for (int i = 0; i < fp.mul_check_limbs; i++) {
            size_t urow;
            if (i == 0)
                urow = base_line({A(lcells[0], id_one), A(rem.limbs_le[0], id_neg_one)}, U(id_neg_one), 0, 0, 0, id_K0);
            else if (i < L)
                urow = base_line({A(lcells[i], id_one), A(rem.limbs_le[i], id_neg_one), A(v_h, id_limb_coeff[1]), A(v_l, id_limb_coeff[0])},
                                 U(id_neg_one), 0, 0, 0, id_K1);
            else
                urow = base_line({A(lcells[i], id_one), A(v_h, id_limb_coeff[1]), A(v_l, id_limb_coeff[0])}, U(id_neg_one), 0, 0, 0, id_K1);
            uint32_t u = mk(0, 4, urow);
            v_h = assign_common();
            v_l = assign_nonleading_limb();
            base_line({A(v_h, id_limb_coeff[2]), A(v_l, id_limb_coeff[1])}, A(u, id_neg_one));
        }
        // native row (integer_chip.rs:205-214)
        base_line({A(a.native, id_zero), A(b.native, id_zero), A(d.native, id_w_native), A(rem.native, id_one)}, none(), id_neg_one);
    }

    // assign_w of an input value (integer_chip.rs:236-258)
    AssignedInteger assign_w(uint32_t input_slot, bool strided = false) {
        push(new_op(H2E_OP_ASSIGN_W, input_slot, strided ? H2E_FLAG_INPUT_STRIDED : 0));
        return shape_assigned(false);
    }
    // reduce (integer_chip.rs:283-373)
    AssignedInteger reduce(const AssignedInteger& a) {
        if (a.times == 1) return a;
        if (!(a.times < (uint64_t)OVERFLOW_LIMIT)) throw std::runtime_error("reduce: times >= overflow_limit");
        H2EOp op = new_op(H2E_OP_REDUCE);
        uint32_t tag = a.vtag;
        bool tag_strided = a.vtag_strided;
        if (tag == H2E_NO_REF && field_hints()) {
            tag = hint_base + hint_count++;
            tag_strided = false;
        }
        if (tag != H2E_NO_REF) {
            op.flags |= H2E_FLAG_HINTED | (tag_strided ? H2E_FLAG_HINT_STRIDED : 0);
            op.imm = tag;
        }
        put_int(op, 0, a);
        push(op);
        AssignedInteger rem = shape_assigned(false);
        rem.vtag = tag;
        rem.vtag_strided = tag_strided;
        uint32_t d = assign_common();
        base_line({A(d, id_w_native), A(rem.native, id_one)}, A(a.native, id_neg_one));
        uint32_t last_v = H2E_NO_REF;
        for (int i = 0; i < fp.reduce_check_limbs; i++) {
            uint32_t v = assign_nonleading_limb();
            base_line({A(d, id_w_limb[i]), A(rem.limbs_le[i], id_one), A(a.limbs_le[i], id_neg_one),
                       last_v != H2E_NO_REF ? A(last_v, id_one) : U(id_zero)},
                      A(v, id_neg_limb_modulus), 0, 0, 0, id_reduce_k[i == 0 ? 0 : 1]);
            last_v = v;
        }
        return rem;
    }
    AssignedInteger conditionally_reduce(const AssignedInteger& a) {  // :375-382
        if (a.times > (uint64_t)REDUCE_THRESHOLD) return reduce(a);
        return a;
    }
    AssignedInteger int_add(const AssignedInteger& a, const AssignedInteger& b) {  // :384-406
        H2EOp op = new_op(H2E_OP_INT_ADD);
        put_int(op, 0, a);
        put_int(op, fp.limbs + 1, b);
        push(op);
        AssignedInteger r;
        for (int i = 0; i < fp.limbs; i++)
            r.limbs_le[i] = mk(0, 4, base_line({A(a.limbs_le[i], id_one), A(b.limbs_le[i], id_one)}, U(id_neg_one)));
        r.native = shape_native_row(r.limbs_le);
        r.times = a.times + b.times;
        take_vtag(r);
        return conditionally_reduce(r);
    }
    AssignedInteger int_sub(const AssignedInteger& a, const AssignedInteger& b) {  // :408-437
        if (b.times >= (uint64_t)OVERFLOW_LIMIT) throw std::runtime_error("int_sub: b.times out of table");
        H2EOp op = new_op(H2E_OP_INT_SUB, (uint32_t)b.times);
        put_int(op, 0, a);
        put_int(op, fp.limbs + 1, b);
        push(op);
        AssignedInteger r;
        for (int i = 0; i < fp.limbs; i++)
            r.limbs_le[i] = mk(0, 4, base_line({A(a.limbs_le[i], id_one), A(b.limbs_le[i], id_neg_one)}, U(id_neg_one), 0, 0, 0,
                                               id_ceil[b.times][i]));
        r.native = shape_native_row(r.limbs_le);
        r.times = a.times + b.times + 1;
        take_vtag(r);
        return conditionally_reduce(r);
    }
    AssignedInteger int_neg(const AssignedInteger& a) {  // :439-464
        if (a.times >= (uint64_t)OVERFLOW_LIMIT) throw std::runtime_error("int_neg: a.times out of table");
        H2EOp op = new_op(H2E_OP_INT_NEG, (uint32_t)a.times);
        put_int(op, 0, a);
        push(op);
        AssignedInteger r;
        for (int i = 0; i < fp.limbs; i++)
            r.limbs_le[i] = mk(0, 4, base_line({A(a.limbs_le[i], id_neg_one)}, U(id_neg_one), 0, 0, 0, id_ceil[a.times][i]));
        r.native = shape_native_row(r.limbs_le);
        r.times = a.times + 1;
        take_vtag(r);
        return conditionally_reduce(r);
    }
    AssignedInteger int_mul(const AssignedInteger& a, const AssignedInteger& b) {  // :466-483
        H2EOp op = new_op(H2E_OP_INT_MUL);
        if (next_vtag == H2E_NO_REF && field_hints()) next_vtag = hint_base + hint_count++;
        if (next_vtag != H2E_NO_REF) {
            op.flags |= H2E_FLAG_HINTED | (in_strand ? H2E_FLAG_HINT_STRIDED : 0);
            op.imm = next_vtag;
        }
        put_int(op, 0, a);
        put_int(op, fp.limbs + 1, b);
        push(op);
        AssignedInteger rem = shape_assigned(false);
        take_vtag(rem);
        AssignedInteger d = shape_assigned(true);
        shape_mul_equation(a, b, d, rem);
        return rem;
    }
    AssignedInteger int_square(const AssignedInteger& a) { return int_mul(a, a); }  // :614-616
    AssignedInteger int_mul_small_constant(const AssignedInteger& a_in, uint64_t b) {  // :618-658
        if (!(b < (uint64_t)REDUCE_THRESHOLD)) throw std::runtime_error("int_mul_small_constant: b >= threshold");
        AssignedInteger a = a_in;
        if (a_in.times * b >= (uint64_t)OVERFLOW_LIMIT) a = reduce(a_in);
        H2EOp op = new_op(H2E_OP_INT_MUL_SMALL, (uint32_t)b);
        put_int(op, 0, a);
        push(op);
        AssignedInteger r;
        for (int i = 0; i < fp.limbs; i++) r.limbs_le[i] = mk(0, 4, base_line({A(a.limbs_le[i], id_small[b])}, U(id_neg_one)));
        r.native = shape_native_row(r.limbs_le);
        r.times = a.times * b;
        take_vtag(r);
        return conditionally_reduce(r);
    }
    // invert rows (base_chip.rs:298-321): returns the condition cell (col 4 of the second row)
    uint32_t shape_is_zero(uint32_t a) {
        size_t r0 = base_line({A(a, id_zero), U(id_zero)}, none(), id_one);
        uint32_t c = mk(0, 1, r0);
        size_t r1 = base_line({A(a, id_zero), U(id_zero)}, A(c, id_one), id_one, 0, 0, id_neg_one);
        if (record_tape) fixups.push_back((uint32_t)(r1 - (in_strand ? strand_off[0] : 0)));  // b = a^-1 filled by the fix-up kernel
        return mk(0, 4, r1);
    }
    // is_int_zero on a reduced operand (integer_chip.rs:540-578)
    AssignedCondition is_int_zero(const AssignedInteger& a_in) {
        AssignedInteger a = reduce(a_in);
        H2EOp op = new_op(H2E_OP_IS_INT_ZERO);
        put_int(op, 0, a);
        push(op);
        // is_pure_zero
        Col cols[H2E_MAX_L];
        for (int i = 0; i < fp.limbs; i++) cols[i] = A(a.limbs_le[i], id_one);
        uint32_t sum = mk(0, 4, base_line(cols, fp.limbs, U(id_neg_one), 0, 0, 0, 0));
        uint32_t is_zero = shape_is_zero(sum);
        // is_pure_w_modulus
        if (a.times != 1) throw std::runtime_error("is_pure_w_modulus: times != 1");
        uint32_t native_diff = mk(0, 4, base_line({A(a.native, id_one)}, U(id_neg_one), 0, 0, 0, id_neg_w_native));
        uint32_t is_eq = shape_is_zero(native_diff);
        for (int i = 0; i < fp.pure_w_check_limbs; i++) {
            uint32_t limb_diff = mk(0, 4, base_line({A(a.limbs_le[i], id_one)}, U(id_neg_one), 0, 0, 0, id_neg_w_limb[i]));
            uint32_t is_limb_eq = shape_is_zero(limb_diff);
            is_eq = mk(0, 4, base_line({A(is_eq, id_zero), A(is_limb_eq, id_zero)}, U(id_neg_one), id_one));
        }
        size_t row = base_line({A(is_zero, id_one), A(is_eq, id_one)}, U(id_neg_one), id_neg_one);
        return AssignedCondition{AssignedValue{mk(0, 4, row)}};
    }
    AssignedCondition is_int_equal(const AssignedInteger& a, const AssignedInteger& b) {  // :47-54
        AssignedInteger diff = int_sub(a, b);
        return is_int_zero(diff);
    }
    // int_div (integer_chip.rs:493-538)
    std::pair<AssignedCondition, AssignedInteger> int_div(const AssignedInteger& a_in, const AssignedInteger& b_in) {
        AssignedInteger b = reduce(b_in);
        AssignedCondition is_b_zero = is_int_zero(b);
        AssignedCondition a_coeff = not_(is_b_zero);
        AssignedInteger ar = reduce(a_in);
        AssignedInteger a;
        {
            H2EOp op = new_op(H2E_OP_MASK_INT);
            put_int(op, 0, ar);
            op.refs[fp.limbs + 1] = a_coeff.v.ref;
            push(op);
            for (int i = 0; i < fp.limbs; i++)
                a.limbs_le[i] = mk(0, 4, base_line({A(ar.limbs_le[i], id_zero), A(a_coeff.v.ref, id_zero)}, U(id_neg_one), id_one));
            a.native = mk(0, 4, base_line({A(ar.native, id_zero), A(a_coeff.v.ref, id_zero)}, U(id_neg_one), id_one));
            a.times = ar.times;
        }
        H2EOp op = new_op(H2E_OP_DIV_CORE);
        next_vtag = H2E_NO_REF;
        if (hint_on && hint_mode == 2 && ecc_block != H2E_NO_REF) {
            op.flags |= H2E_FLAG_HINTED | (in_strand ? H2E_FLAG_HINT_STRIDED : 0);
            op.imm = ecc_block + H2E_HINT_LAMBDA;
        } else if (hint_on && (hint_mode == 1 || field_hints())) {
            op.flags |= H2E_FLAG_HINTED | (in_strand ? H2E_FLAG_HINT_STRIDED : 0);
            op.imm = hint_base + hint_count++;
        }
        put_int(op, 0, b);
        put_int(op, fp.limbs + 1, a);
        push(op);
        AssignedInteger c = shape_assigned(false);
        if (op.flags & H2E_FLAG_HINTED) {
            c.vtag = op.imm;
            c.vtag_strided = (op.flags & H2E_FLAG_HINT_STRIDED) != 0;
        }
        AssignedInteger d = shape_assigned(true);
        shape_mul_equation(b, c, d, a);
        return std::make_pair(is_b_zero, c);
    }
    // assign_int_constant (integer_chip.rs:580-598) for a canonical W value known at record time
    AssignedInteger assign_int_constant(const HBig& w_value) {
        push(new_op(H2E_OP_CONST_INT, pool_w(w_value)));
        AssignedInteger r;
        for (int i = 0; i < fp.limbs; i++) {
            uint32_t cid = emit_shape ? intern(fp.fr(w_value.shr(i * LIMB_BITS).low_bits(LIMB_BITS))) : 1;
            r.limbs_le[i] = mk(0, 0, base_line({U(id_neg_one)}, none(), 0, 0, 0, cid));
        }
        uint32_t cid = emit_shape ? intern(fp.fr(w_value)) : 1;
        r.native = mk(0, 0, base_line({U(id_neg_one)}, none(), 0, 0, 0, cid));
        r.times = 1;
        return r;
    }
    // assign_int_constant whose value is an instance input (e.g. the G2 point of a pairing check): the
    // `constant` fixed cells of these rows are reported as FixedPatch entries instead of dictionary ids.
    AssignedInteger assign_int_constant_input(uint32_t input_slot) {
        push(new_op(H2E_OP_CONST_INT_INPUT, input_slot));
        AssignedInteger r;
        for (int i = 0; i <= fp.limbs; i++) {
            size_t row = base_line({U(id_neg_one)}, none(), 0, 0, 0, 0);
            if (emit_shape) fixed_patches.push_back(FixedPatch{(uint32_t)row, 8, input_slot, i < fp.limbs ? i : -1});
            if (i < fp.limbs) r.limbs_le[i] = mk(0, 0, row);
            else r.native = mk(0, 0, row);
        }
        r.times = 1;
        return r;
    }
    // assert_int_equal (integer_chip.rs:600-612)
    void assert_int_equal(const AssignedInteger& a, const AssignedInteger& b) {
        AssignedInteger diff = int_sub(a, b);
        diff = reduce(diff);
        H2EOp op = new_op(H2E_OP_SUM_LIMBS);
        for (int i = 0; i < fp.limbs; i++) op.refs[i] = diff.limbs_le[i];
        push(op);
        Col cols[H2E_MAX_L];
        for (int i = 0; i < fp.limbs; i++) cols[i] = A(diff.limbs_le[i], id_one);
        uint32_t sum = mk(0, 4, base_line(cols, fp.limbs, U(id_neg_one), 0, 0, 0, 0));
        assert_constant(AssignedValue{sum}, 0);
    }
    // int_unsafe_invert (integer_chip.rs:485-491)
    AssignedInteger int_unsafe_invert(const AssignedInteger& x) {
        AssignedInteger one = assign_int_constant(HBig(1));
        auto r = int_div(one, x);
        assert_false(r.first);
        return r.second;
    }
    // bisec_int (integer_chip.rs:660-681)
    AssignedInteger bisec_int(const AssignedCondition& cond, const AssignedInteger& a, const AssignedInteger& b) {
        return bisec_int_limbs(cond, a, b, fp.limbs);
    }
    // bisec_int of integers of `limbs` limbs, whatever the current field is: the rows are base-chip rows only, so the
    // other integer context of a GeneralScalarEccContext can call it from inside a fork of this one (ecc_bisec_scalar next
    // to ecc_bisec_to_non_zero_point in msm_unsafe's loop, ecc_chip.rs:386-391).  op.imm = limbs.
    AssignedInteger bisec_int_limbs(const AssignedCondition& cond, const AssignedInteger& a, const AssignedInteger& b, int limbs) {
        H2EOp op = new_op(H2E_OP_BISEC_INT, (uint32_t)limbs);
        op.refs[0] = cond.v.ref;
        for (int i = 0; i < limbs; i++) {
            op.refs[1 + i] = a.limbs_le[i];
            op.refs[limbs + 2 + i] = b.limbs_le[i];
        }
        op.refs[1 + limbs] = a.native;
        op.refs[2 * limbs + 2] = b.native;
        push(op);
        AssignedInteger r;
        for (int i = 0; i < limbs; i++) r.limbs_le[i] = mk(0, 4, bisec_row(cond.v.ref, a.limbs_le[i], b.limbs_le[i]));
        r.native = mk(0, 4, bisec_row(cond.v.ref, a.native, b.native));
        r.times = std::max(a.times, b.times);
        return r;
    }
    // One limb of GeneralScalarEccContext::decompose_scalar::<1> (general_scalar_ecc_chip.rs:107-130): per bit j of the
    // limb (LSB first) assign_bit(b_j), then [rest * -1, b_j * 1 | v * 2] with v = (rest - b_j) / 2 becoming the next
    // `rest`; finally assert_constant(rest, 0).  Returns the bit cells in the order they are pushed (LSB first).
    std::vector<AssignedCondition> decompose_limb(uint32_t limb_ref, int bits) {
        H2EOp op = new_op(H2E_OP_DECOMPOSE_LIMB, (uint32_t)bits);
        op.refs[0] = limb_ref;
        push(op);
        std::vector<AssignedCondition> out;
        uint32_t rest = limb_ref;
        for (int j = 0; j < bits; j++) {
            size_t brow = base_line({U(id_one), U(id_zero)}, none(), id_neg_one);      // assign_bit (base_chip.rs:357-367)
            uint32_t b = mk(0, 0, brow);
            size_t row = base_line({A(rest, id_neg_one), A(b, id_one)}, U(id_two));
            rest = mk(0, 4, row);
            out.push_back(AssignedCondition{AssignedValue{b}});
        }
        base_line({A(rest, id_neg_one)}, none(), 0, 0, 0, id_zero);                      // assert_constant(rest, 0)
        return out;
    }
};

}  // namespace h2e
