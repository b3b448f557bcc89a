// Witness tape: the static (shape-only) op stream the host recorder emits and the HIP engine replays.
//
// The reference's witness program is static: every branch in L1-L4 depends only on shape (`times`,
// limb counts, group sizes, NAF digits), never on field values, except the *value* picked by
// `pick_candidate_non_zero` (src/circuit/ecc_chip.rs:935-953) and abort flags.  So one recording of
// the chip-level calls (at IntegerChipOps granularity, src/circuit/integer_chip.rs:15-70) is replayed
// for any number of instances; forked sub-contexts (`ParallelClone`, src/circuit/ecc_chip.rs:64-77)
// become "strands": one recording replayed at row offsets strand*delta.
#pragma once
#include <stdint.h>

// ---- cell reference (32 bit) -------------------------------------------------------------------
//  [31:30] region: 0 base, 1 range, 2 select, 3 = per-strand parameter (index in [25:0])
//  [29:27] column
//  [26]    1 = row is relative to the strand's starting offset of that region
//  [25:0]  row
#define H2E_REF_REGION(r) ((r) >> 30)
#define H2E_REF_COL(r) (((r) >> 27) & 7u)
#define H2E_REF_REL(r) (((r) >> 26) & 1u)
#define H2E_REF_ROW(r) ((r)&0x3ffffffu)
#define H2E_MAKE_REF(region, col, rel, row) \
    (((uint32_t)(region) << 30) | ((uint32_t)(col) << 27) | ((uint32_t)(rel) << 26) | ((uint32_t)(row)&0x3ffffffu))
#define H2E_REGION_PARAM 3u
#define H2E_NO_REF 0xffffffffu

enum H2EOpcode {
    H2E_OP_NOP = 0,
    // values entering from the instance input vector (imm = input slot, + strand*input_stride if flag)
    H2E_OP_ASSIGN_W,          // assign_w(input)                    integer_chip.rs:236-258
    H2E_OP_ASSIGN,            // base_chip assign(input Fr)          base_chip.rs:351-355
    H2E_OP_ASSIGN_BIT,        // assign_bit(input bit)               base_chip.rs:357-367
    H2E_OP_CONST_INT,         // assign_int_constant(pool const)     integer_chip.rs:580-598
    H2E_OP_CONST_INT_INPUT,   // assign_int_constant(input W value)  (G2 / expected-result constants)
    H2E_OP_CONST,             // assign_constant(pool Fr const)      base_chip.rs:344-349
    // integer chip
    H2E_OP_INT_ADD,           // integer_chip.rs:384-406 (without the conditional reduce)
    H2E_OP_INT_SUB,           // :408-437   imm = b.times
    H2E_OP_INT_NEG,           // :439-464   imm = a.times
    H2E_OP_INT_MUL_SMALL,     // :618-658   imm = k
    H2E_OP_INT_MUL,           // :466-483 + :73-215
    H2E_OP_REDUCE,            // :283-373
    H2E_OP_IS_INT_ZERO,       // :540-578 on an already reduced operand (is_pure_zero | is_pure_w_modulus, or)
    H2E_OP_NOT,               // base_chip.rs:398-403
    H2E_OP_MASK_INT,          // int_div's a' = a * not(is_b_zero), limb-wise + native   :511-520
    H2E_OP_DIV_CORE,          // int_div's c,d hints + assign_w/assign_d + mul equation  :522-535
    H2E_OP_BISEC_INT,         // :660-681   imm = limbs (0: the kernel's field; a GeneralScalarEccContext bisects scalars
                              //            of its other integer context inside a fork of the base field)
    H2E_OP_SUM_LIMBS,         // sum_with_constant(limbs, 1) of assert_int_equal  :607-610
    // base chip rows
    H2E_OP_ASSERT_CONST,      // assert_constant(x, imm in {0,1})    base_chip.rs:375-379 ; flags a status on mismatch
    H2E_OP_BISEC,             // base_chip.rs:574-604
    H2E_OP_AND,               // :392-396
    H2E_OP_OR,                // :428-439
    H2E_OP_XNOR,              // :455-467
    H2E_OP_DECOMPOSE_NATIVE,  // native_scalar_ecc_chip.rs:97-171 (WINDOW_SIZE = 1), imm = NUM_BITS
    H2E_OP_PICK_INDEX,        // pick_candidate_non_zero's index row(s)  ecc_chip.rs:941-948
    // select chip
    H2E_OP_CACHE_INT,         // assign_cache_integer      ecc_chip.rs:734-751; imm = cells to cache (0: L + 1, an integer)
    H2E_OP_SELECT_POINT,      // assign_selected_point_non_zero  ecc_chip.rs:955-967 (value picked by index cell);
                              // flags bits 8-15 = cells per candidate (0: 2 (L + 1)): assign_selected_point :790-812 selects
                              // x, y, z, curvature (3 L + 5 cells; at most 16)
    // general (non-native) scalars
    H2E_OP_DECOMPOSE_LIMB,    // one limb of decompose_scalar::<1>  general_scalar_ecc_chip.rs:107-130, imm = limb bits
    H2E_OP_SHIFT_ADD,         // sum_with_constant([(a, 1), (b, 2^108)])  of ecc_encode  ecc_chip.rs:718-731
    H2E_OP_COUNT
};

// status bits (per instance), or-ed by the engine
#define H2E_STATUS_OK 0u
#define H2E_STATUS_ASSERT_FAILED 1u        // an assert_constant / assert_true / assert_false would have panicked
#define H2E_STATUS_RETRY_ADD_SAME_OR_NEG 2u  // UnsafeError::AddSameOrNegPoint  ecc_chip.rs:853-857
#define H2E_STATUS_RETRY_ADD_IDENTITY 4u     // UnsafeError::AddIdentity        ecc_chip.rs:877-881
#define H2E_STATUS_ARITH 8u                // an internal exactness check failed (u % 2^108 != 0, ...)

// flags in H2EOp.flags
#define H2E_FLAG_INPUT_STRIDED 1u   // input slot = imm + strand * input_stride
#define H2E_FLAG_UNSAFE_ADD 2u      // ASSERT_CONST failure reports RETRY_ADD_SAME_OR_NEG
#define H2E_FLAG_UNSAFE_DBL 4u      // ASSERT_CONST failure reports RETRY_ADD_IDENTITY
#define H2E_FLAG_HINTED 8u          // DIV_CORE: quotient c = a/b comes from the hint buffer, slot = imm (+ strand*hint_stride)
                                    // INT_MUL / REDUCE: the values-only replay takes the (canonical) result from that
                                    // slot; the full expansion computes it and checks the hint (H2E_STATUS_ARITH)
#define H2E_FLAG_HINT_STRIDED 16u
#define H2E_FLAG_LOCAL_RESULT 32u   // the result is only read inside the op's own sub-range: the values-only replay
                                    // keeps it in LDS and does not store it (set by the recorder's liveness pass)

#define H2E_FLAG_VALUES_SKIP 64u    // nothing the values-only replay has to produce depends on this op (host DCE pass)
#define H2E_FLAG_PRESELECTED 128u   // PICK_INDEX / SELECT_POINT of an MSM window: a select pre-kernel (H2E_PRE_MSM_SELECT) has
                                    // already picked the candidate into the selection buffer, entry refs[1] (+ strand * stride),
                                    // so the value chain has no data-dependent addresses; the expansion ignores the flag

// "full value hints" of an MSM chain: every ecc_add_unsafe / ecc_double_unsafe owns a block of 8 hint slots with the
// canonical W value of each mul-like result of src/circuit/ecc_chip.rs:814-882 (a = first argument, c = result):
#define H2E_ECC_HINT_SLOTS 8u
#define H2E_HINT_LAMBDA 0u    // int_div quotient
#define H2E_HINT_LAMBDA2 1u   // int_square(lambda)
#define H2E_HINT_XC 2u        // c.x   (a later reduce of cx)
#define H2E_HINT_YC 3u        // c.y
#define H2E_HINT_T2L 4u       // (a.x - c.x) * lambda
#define H2E_HINT_T2 5u        // a.x - c.x
#define H2E_HINT_AUX0 6u      // add: a.x - b.x (reduce inside int_div)   double: a.x^2
#define H2E_HINT_AUX1 7u      // double: 2 a.y ; scratch of the finalize kernel otherwise
// ecc op kinds of a chain (2 bits each in H2EPreKernel.pattern), "prev" = the previous result of the chain
#define H2E_ECC_DBL 0u            // a = b = prev
#define H2E_ECC_ADD_EXT_PREV 1u   // a = external point, b = prev
#define H2E_ECC_ADD_PREV_EXT 2u   // a = prev, b = external point

#define H2E_OP_MAX_REFS 11
typedef struct H2EOp {
    uint16_t opcode;
    uint16_t flags;
    uint32_t imm;         // opcode specific
    uint32_t base_row;    // first base row written (strand relative if the tape is a strand tape)
    uint32_t range_row;   // first range row written
    uint32_t select_row;  // first select row written
    uint32_t refs[H2E_OP_MAX_REFS];
} H2EOp;  // 64 bytes

// Field-pair constants the engine needs (derived on the host from RangeInfo, src/range_info.rs:77-184).
// All multiword integers little-endian 64-bit words.
#define H2E_MAX_L 4
#define H2E_W_WORDS_MAX 6
typedef struct H2EFieldConsts {
    uint32_t limbs;                 // L (3 or 4)
    uint32_t w_words;               // 4 or 6
    uint32_t w_bits;                // bit length of w (k)
    uint32_t w_ceil_bits;
    uint32_t d_bits;
    uint32_t w_lead_bits, d_lead_bits;     // bits of the leading limb of a W element / of a quotient
    uint32_t mul_check_limbs, reduce_check_limbs, pure_w_check_limbs;
    uint32_t barrett_s;             // X < 2^s for the w-Barrett (= 2*w_ceil_bits + 2*overflow_bits)
    uint32_t input_bytes;           // 32 or 48: size of a W value in the input vector
    uint64_t w[H2E_W_WORDS_MAX];    // modulus of W
    uint64_t w_mu[8];               // floor(2^s / w)
    uint64_t w_limbs[H2E_MAX_L][2]; // limbs of w (108 bit)
    uint64_t n[4];                  // bn256 Fr modulus
    uint64_t n_mu[5];               // floor(2^512 / n)
    uint64_t w_native[4];           // w mod n
    uint64_t ceil_limbs[64][H2E_MAX_L][2];  // find_w_modulus_of_ceil_times(t) limbs (range_info.rs:334-359)
    uint64_t ceil_native[64][4];            // their composition mod n
    // Montgomery constants (R = 2^(64*words)) for the value-predictor kernels and the inverse fix-up
    uint64_t w_minv;                        // -w^-1 mod 2^64
    uint64_t w_r2[H2E_W_WORDS_MAX];         // R^2 mod w
    uint64_t w_r1[H2E_W_WORDS_MAX];         // R mod w
    uint64_t n_minv;                        // -n^-1 mod 2^64
    uint64_t n_r2[4];                       // R^2 mod n
    uint64_t n_r1[4];                       // R mod n
    // digit-parallel field chain (engine.hip h2e_field_chain_digits): the 32-bit "digits" beta_j = 2^44 + delta_j of a multiple
    // of w, delta = (-2^44 * sum_j 2^(32 j)) mod w - added to the columns of a signed linear combination they keep every
    // column positive without changing the value mod w
    uint64_t lin_bias[2 * H2E_W_WORDS_MAX];
} H2EFieldConsts;

struct H2EVRec;
// One launch: a tape replayed by n_instances * n_strands lanes.
typedef struct H2ELaunch {
    const H2EOp* tape;
    uint32_t n_ops;
    uint32_t n_strands;           // strands per instance (1 for the main context)
    uint32_t strand_base0, strand_range0, strand_select0;   // offsets of strand 0
    uint32_t delta_base, delta_range, delta_select;         // per-strand Offset (ecc_chip.rs:36-41)
    uint32_t input_stride;        // input slots per strand (for H2E_FLAG_INPUT_STRIDED)
    uint32_t n_params;            // parameter refs per strand
    const uint32_t* params;       // [n_strands][n_params]
    const uint32_t* aux;          // candidate tables etc.
    const uint64_t* const_pool;   // 32-byte (Fr) or w_words*8-byte (W) constants, word indexed
    uint32_t hint_stride;         // hint slots per strand (for H2E_FLAG_HINT_STRIDED)
    uint32_t n_fixups;            // is_zero inverse cells per strand filled by the fix-up kernel after this launch
    const uint32_t* fixups;       // [n_fixups] strand-relative base rows: x = (row, col 0), inverse -> (row, col 1)
    // Sub-ranges: after a values-only replay of the whole tape has put every *result* cell in place, the full
    // expansion of op ranges [sub[k], sub[k+1]) is independent for different k and runs as separate lanes.
    uint32_t rel_refs;            // 1 = cells created by this tape are strand-relative refs (fork segment)
    uint32_t n_sub;               // 0/1 = the whole tape per lane
    const uint32_t* sub;          // [n_sub + 1] op indices relative to `tape`
    const struct H2EVRec* vtape;  // compiled values-only replay (cut segments only): the program's record array
    const uint32_t* vpieces;      // [n_vpieces][2] first / end record of each independent piece of this segment's replay
    uint32_t n_vpieces;
    uint32_t v_int_slots, v_units; // LDS sizing of the replay kernel: integer slots and 16-byte staging units per lane
    uint32_t sel_stride;          // selection-buffer entries per strand (H2E_FLAG_PRESELECTED)
    uint32_t x_blocks;            // expansion launched in its persistent form: the 64-lane blocks the launch's workgroups share (0: one each)
    // level-parallel replay (h2e_capi.cpp compile_replay): H2E_LEVEL_WAVES x 64 records per round; thread t of the
    // workgroup runs record 64 * H2E_LEVEL_WAVES * round + t; l_steps = rounds
    const struct H2EVRec* lrecs;
    const uint32_t* lrefs;        // cell refs of global integer operands (L + 1 each)
    uint32_t l_steps, l_slots;
    uint32_t field_pair;          // the W field of this segment's integer ops (host side: which kernel instantiation)
    uint32_t slot_words;          // words per input slot (the program's: 6 for a bls12_381 Fq program even in its Fr segments)
    uint32_t l_pair;              // level-parallel replay: 1 = two instances per workgroup (lanes 0-31 / 32-63), steps of 32 ops;
                                  // 2 = wave mode: one wave per instance, rounds of up to 64 compact records (lrounds / l_recs)
    const uint32_t* lrounds;      // wave mode: per round (first record, count | kind << 8); kind 0 = light ops of mixed opcodes
    uint32_t l_recs;              // wave mode: records incl. padding (a multiple of H2E_WCHUNK)
    // hint store (field_chain.hpp HintStore): replaces the values-only replay of a segment whose mul-like results all have
    // hints - one lane per (store op, instance)
    const uint32_t* s_words;      // store op records
    const uint32_t* s_offsets;    // [n_sops] first word of each record
    const uint64_t* s_ktab;       // K constants, (2 L + 4) words each
    uint32_t n_sops;
    const uint32_t* s_ext;        // extension table of the store records' leaves (H2E_SX_WORDS words per entry, engine.hip hs_leaf)
    // stream digest of the run (h2e.h h2e_run_digest): [dg_shards][3][n_instances][4] words the expansion / fix-up kernels add to
    // (a workgroup adds to shard blockIdx.x mod dg_shards - 3.3 M lanes adding to the 768 words of 64 instances would queue up
    // behind a handful of L2 channels; h2e_digest_reduce sums the shards at the end of the run); NULL = off
    uint64_t* dg_out;
    uint32_t dg_shards;           // a power of two
    // packed expansion (batches smaller than half a wave, engine.hip h2e_run_tape_packed): the order its waves take the sub-ranges
    // in.  Sub-ranges with the same opcode sequence share a wave - its groups then never wait for each other's ops - and the
    // heaviest waves come first; ~0u = an empty group slot (a class's last wave is padded).  Host side: one table per group count
    // G = 2 << k (k = 0..4) behind each other; the launcher hands the kernel the one it runs with in pk_order.
    const uint32_t* pk_order;     // device
    uint32_t pk_off[5], pk_waves[5];   // first entry / waves of table k (pk_waves[k] = 0: none - tape order)
    uint32_t pk_n_sub;            // the n_sub the tables were built for (a launch over part of the sub-ranges runs in tape order)
    // column emission (h2e.h h2e_run_columns): halo2's advice columns straight out of the expansion.  col[region] = instance 0's
    // per-instance column-major array [col][col_rows[region]][4 words] of the region (base, range, select), col_stride words to the
    // next instance's; NULL = off.  The batch-interleaved arrays stay the working copy operands are read from.
    uint64_t* col[3];
    uint64_t col_stride[3];
    uint32_t col_rows[3];
    uint32_t col_form;            // 0 canonical, 1 Montgomery
} H2ELaunch;
enum H2EStoreKind { H2E_S_W = 1, H2E_S_LIN = 2, H2E_S_FE = 3, H2E_S_CONST = 4, H2E_S_FULL = 5 };
// leaves that do not fit a term word (kind 3: index into H2ELaunch::s_ext): strand-strided hint / selection / input slots and
// integers read from cells that were written before the launch
enum H2EStoreExt { H2E_SX_HINT = 0, H2E_SX_SEL = 1, H2E_SX_CELLS = 2, H2E_SX_INPUT = 3 };
#define H2E_SX_WORDS 8u

// ---- compiled values-only replay ("V-tape") ----------------------------------------------------
// The values-only replay of a cut segment does not interpret the witness tape: the host compiles it (once per
// shape) into a stream of 32-byte records that holds only the ops whose results something depends on, with every
// operand already resolved to either an LDS slot (static allocation by live range, spilled values go through
// their cells) or a cell reference.  An op = one header record + n_ext extension records of 8 words.
//   w[0] = vopcode | vflags << 8 | dst slot << 16 | n_ext << 24        (dst 0xff: result not kept in a slot)
//   w[1] = imm            (times / k / hint slot or staging unit / aux offset)
//   w[2..4] = src0..2     int / fe in a slot: slot; staged int / fe: first staging unit; global fe: the cell ref;
//                         global int: word offset into the op's extension records of its L+1 cell refs
//   w[5] = base row, w[6] = range (SELECT_POINT: select) row of the result cells (strand relative like the tape's)
//   w[7] = operand kinds, 3 bits per source (H2E_VSRC_*); SELECT_POINT: bits 16..23 second dst slot
// Inputs that come from memory (hints, cells written by other kernels) are *staged*: each piece of the replay starts
// with H2E_V_GATHER records listing them in 16-byte units; the kernel issues them as asynchronous global->LDS loads
// and waits once (H2E_V_GATHER_WAIT), so the dependent chain itself never waits for memory.
//   gather record: w[0] = H2E_V_GATHER | n << 8 (n <= 3 entries), w[1] = first staging unit,
//                  w[2 + 2e] = kind (0 cell, 1 hint, 2 selection entry) | 16-byte piece << 4 | strided << 8,
//                  w[3 + 2e] = cell ref / hint slot / selection entry
typedef struct H2EVRec {
    uint32_t w[8];
} H2EVRec;
#define H2E_VCHUNK 128u   // records per LDS chunk; an op never straddles a chunk boundary (host pads with H2E_V_NOP)
enum H2EVOpcode {
    H2E_V_NOP = 0,
    H2E_V_HINT,          // mul-like result := hint[imm]               (hinted INT_MUL / REDUCE / DIV_CORE)
    H2E_V_MUL, H2E_V_REDUCE, H2E_V_DIV,
    H2E_V_ADD, H2E_V_SUB, H2E_V_NEG, H2E_V_MUL_SMALL, H2E_V_MASK, H2E_V_BISEC_INT,
    H2E_V_IS_ZERO, H2E_V_NOT, H2E_V_AND, H2E_V_OR, H2E_V_XNOR, H2E_V_PICK_INDEX,
    H2E_V_SELECT_POINT,
    H2E_V_FULL,          // run the tape op held in the 2 extension records as it is (its rows are its results)
    H2E_V_GATHER, H2E_V_GATHER_WAIT,
    H2E_V_LOAD_SEL,      // both coordinates of a pre-selected point: src0 = staging unit (or selection entry), two dst slots
    H2E_V_CONST          // assign_int_constant of a pool constant (imm = pool word offset): limbs in column 0 of base_row + i
};
#define H2E_VFLAG_STORE 1u          // the result is also written to its cells
#define H2E_VFLAG_HINT_STRIDED 2u
#define H2E_VFLAG_STAGED 4u         // H2E_V_HINT: imm is a staging unit
#define H2E_VFLAG_FENCE 8u          // level-parallel replay: this round holds an H2E_V_FULL op - fence before the barrier
#define H2E_VFLAG_MIXED 16u         // level-parallel replay: a step of light ops (additions, selections, conditions) of any
                                    // mix of opcodes - every lane runs its own record's opcode
#define H2E_VSRC_NONE 0u
#define H2E_VSRC_INT_SLOT 1u
#define H2E_VSRC_FE_SLOT 2u
#define H2E_VSRC_GLOBAL 3u
#define H2E_VSRC_STAGE 4u
#define H2E_V_NO_SLOT 0xffu
#define H2E_LEVEL_WAVES 4u   // waves of a level-parallel replay workgroup (they share one instance's value slots)
#define H2E_WCHUNK 256u      // wave mode: records per LDS chunk buffer; a round never straddles a chunk (host pads with H2E_V_NOP)
#define H2E_DP_CHUNKS 4u     // digit-parallel field chain: record chunks the kernel keeps in LDS (a ring, 16 KB each); the host's
                             // LDS budget check (field_chain.hpp) and the launcher (engine.hip) size from this one constant

// ---- value-predictor ("V") kernels for the MSM ---------------------------------------------------
// They run native Montgomery / Jacobian arithmetic over the same inputs, write numerator/denominator pairs of
// every lambda = dy/dx the ecc_add_unsafe / ecc_double_unsafe chain will divide (src/circuit/ecc_chip.rs:840-882),
// and one batch inversion turns them into hints for H2E_OP_DIV_CORE.
enum H2EPreKind { H2E_PRE_MSM_CANDIDATES = 1, H2E_PRE_MSM_WINDOWS = 2, H2E_PRE_MSM_TAIL = 3, H2E_PRE_MSM_SELECT = 4,
                  H2E_PRE_FIELD_CHAIN = 5 };   // field_chain.hpp: residues mod w in Montgomery form, one wave per instance
// selection buffer: per (strand, group) the picked candidate, x then y, canonical, H2E_W_WORDS_MAX words each
#define H2E_SEL_WORDS (2 * H2E_W_WORDS_MAX)
// selection buffer entry = 4 value slots: x, y canonical (what the replay / expansion read), x, y in Montgomery form (what the
// windows' scan predictor multiplies: it walks every candidate twice and would convert it both times)
#define H2E_SEL_SLOTS 4u
typedef struct H2EPreKernel {
    uint32_t kind;
    uint32_t n_lanes;        // lanes per instance (groups / windows / 1)
    uint32_t hint_base;      // first hint slot written
    uint32_t hints_per_lane; // consecutive hint slots per lane
    uint32_t args_begin;     // index into the pre-kernel args array (uint32)
    uint32_t n_params;       // parameter refs per lane (reuses the X segment's parameter table)
    uint32_t params_begin;
    uint32_t scratch_begin;  // per-instance Jacobian scratch: first slot (96-byte slots)
    // full value hints (0 = quotient hints only): ecc ops per lane, hints_per_lane = 8 * (ecc_ops + 1); the op kinds
    // repeat with period pattern_len (2 bits each, first op in the low bits)
    uint32_t ecc_ops;
    uint32_t pattern_len;
    uint32_t pattern;
    uint32_t sel_begin;      // first entry of this kernel's strands in the selection buffer (SELECT writes, WINDOWS reads)
    uint32_t used_slots;     // full value hints: bit k set = slot k of the ecc blocks is read by someone (host, after the DCE pass)
    uint32_t scan_begin;     // WINDOWS / TAIL: first Jacobian scratch slot of the chain's scan (sizes: H2E_WIN_SCAN_SLOTS / H2E_TAIL_SCAN_SLOTS)
    // FIELD_CHAIN: the program sits in the pre-kernel args array (32-byte aligned): records of 8 words from word f_recs
    // (f_n_recs of them, a multiple of H2E_WCHUNK), (first record, count | kind << 8) per round from word f_rounds;
    // hint slots [hint_base, hint_base + hints_per_lane) are turned into canonical values by the finalize kernel
    uint32_t f_recs, f_n_recs, f_rounds, f_n_rounds, f_slots;
    uint32_t f_n_load_rounds;   // the first rounds: loads of inputs / constants (a loop of their own in the kernel)
    uint32_t f_sinks, f_sink_words, f_n_sinks;   // hint-only linear combinations computed after the chain (h2e_field_sinks): word index of the
                                                 // per-sink offsets / of the sink records in the args array (field_chain.hpp FieldChain), count
    uint32_t hint2_base, hints2_per_lane;   // FIELD_CHAIN: a second range of hint slots to finalize - the slots taken when the program was compiled
                                            // (conditions, sink terms, values later segments import); with several segments another segment's
                                            // recorded slots lie between the two ranges
    uint32_t f_mode;            // 0 = one lane per record, records of 8 words (h2e_field_chain); 1 = one 16-lane row per record, records of
                                // 16 words = up to 14 terms per linear combination (h2e_field_chain_digits)
    uint32_t* f_started;        // digit chain: every workgroup adds 1 here when it starts (the gate in front of the expansion that became
                                // ready with it, h2e_engine_gate); NULL: nobody waits
} H2EPreKernel;
// field chain record opcodes (field_chain.hpp FieldCompiler::F_*)
enum H2EFieldOp { H2E_F_NOP = 0, H2E_F_LIN, H2E_F_MUL, H2E_F_DIV, H2E_F_ISZERO, H2E_F_NOT, H2E_F_AND, H2E_F_OR, H2E_F_XNOR, H2E_F_SELECT,
                  H2E_F_INPUT_W, H2E_F_INPUT_FE, H2E_F_CONST_W, H2E_F_CONST_FE,
                  H2E_F_RESERVED_14, H2E_F_CONT };   // CONT: the second record of a long combination (15 .. 28 terms), behind the round's rows
#define H2E_F_FROM_HINTS 0x100u   // flag in word 0 of an H2E_F_INPUT_W record: word 2 is a hint slot an earlier segment's chain left the value in
#define H2E_DP_DIV_SCRATCH 320u   // digit chain: LDS behind the value slots for the state of the loader wave's inversion (4 NL + N + 1 64-bit words, NL = 7, N = 6: 280 B)
#define H2E_F_MAX_TERMS 6        // 8-word records
#define H2E_F_MAX_TERMS_WIDE 14   // 16-word records
// The MSM chains are walked as scans (engine.hip "scan predictors"): a window's sum over its groups in H2E_WIN_CHUNKS
// chunks (chunk sums -> offsets -> the real additions of every chunk in parallel), the tail's accumulation
// acc <- 2 acc + line_w [- r2] in chunks of H2E_TAIL_CHUNK windows (local Horner sums B, the doubling chain D of the
// chunk's start value A, acc_w = D_w + B_w).  Scratch slots per instance:
#define H2E_WIN_CHUNKS 8u
#define H2E_TAIL_CHUNK 16u
#define H2E_WIN_SCAN_SLOTS(windows) ((windows) * 2u * H2E_WIN_CHUNKS)                                   /* S_c, O_c per window */
#define H2E_TAIL_SCAN_SLOTS(windows) (2u * (windows) + ((windows) + H2E_TAIL_CHUNK - 1u) / H2E_TAIL_CHUNK + 1u)  /* B_w, D_w, A_c, (r1 in Montgomery form | fallback flag) */
