/* h2e — C ABI of the MI355X witness-generation engine for the halo2ecc-s hot path.
 *
 * Drop-in boundary.  The reference has no FFI layer; its mechanism for running a
 * sub-computation elsewhere and splicing the result back is `ParallelClone` (fork a context at a row
 * offset, let it write its disjoint rows, merge: src/circuit/ecc_chip.rs:64-77, used at :289-352).
 * The engine plugs in at that seam: a *program* is the recorded shape of a chip-level computation
 * (rows, fixed cells, permutations, heights = everything in `Records` that does not depend on field
 * values, src/context.rs:241-301), and `h2e_run` fills the advice values of N instances of it on the
 * GPU.  INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions
 *  - all multi-word integers are little-endian arrays of uint64_t; advice cells are canonical bn256-Fr
 *    values (what `field_to_bn` sees, src/utils.rs:4-8), 4 words each;
 *  - advice arrays are *batch-interleaved*: rows and columns are the reference's
 *    (`Vec<[(Option<N>, bool); COLS]>`, src/context.rs:243-251; COLS = 5 base, 3 range, 2 select), a cell is two
 *    16-byte halves, and the n_instances instances of a run are the minor dimension:
 *        array[row][col][half][instance][2 words]      (n_instances * rows * COLS * 4 words in all)
 *    i.e. word k of cell (row, col) of instance i is at
 *        ((row * COLS + col) * 2 + k / 2) * 2 * n_instances + 2 * i + k % 2.
 *    With n_instances == 1 this is exactly the reference's row-major [rows][COLS][4].  Why: the 64 lanes of a
 *    wavefront are 64 instances at the same cell, so every store the engine issues is a contiguous 1 KB run and
 *    only assigned cells cost HBM bandwidth.  The engine writes assigned cells only (the assigned / permute
 *    flags are shape artefacts, h2e_program_shape): cells the shape leaves unassigned keep whatever the caller's
 *    buffer held - zero-fill the arrays, or read them through h2e_export, which masks with the flags;
 *  - h2e_export turns a batch-interleaved array into one array per instance, row-major (Records) or
 *    column-major (halo2 advice columns), canonical or Montgomery-form cells;
 *  - inputs: [n_instances][n_input_slots][slot_words] words; a slot holds one W value (canonical) or one
 *    Fr value / flag in its first 4 words;
 *  - device pointers are plain `void*` from any allocator (hipMalloc, torch); the engine never frees
 *    caller memory; `stream` is a hipStream_t passed as void*;
 *  - every function returns 0 on success, a negative H2E_ERR_* otherwise; h2e_last_error() has text.
 */
#ifndef H2E_H
#define H2E_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define H2E_ERR_INVALID (-1)
#define H2E_ERR_HIP (-2)
#define H2E_ERR_SHAPE (-3)

/* field pairs: RangeInfo::<W, N> instances the reference constructs (src/range_info.rs:362-387) */
#define H2E_FIELD_BN256_FQ 0       /* bn256 Fq over bn256 Fr: 3 limbs */
#define H2E_FIELD_BLS12_381_FQ 1   /* bls12_381 Fq over bn256 Fr: 4 limbs */
#define H2E_FIELD_BLS12_381_FR 2   /* bls12_381 Fr over bn256 Fr: 3 limbs */

/* per-instance status bits (would-be panics / UnsafeError of the reference, src/circuit/ecc_chip.rs:23-34) */
#define H2E_ST_OK 0u
#define H2E_ST_ASSERT_FAILED 1u
#define H2E_ST_RETRY_ADD_SAME_OR_NEG_POINT 2u
#define H2E_ST_RETRY_ADD_IDENTITY 4u
#define H2E_ST_ARITH 8u
#define H2E_ST_TEST_HOOK 0x80u   /* the run was made with a test hook that leaves rows out (h2e_ctx_set_option) */

typedef struct h2e_ctx h2e_ctx;          /* device + constant tables; one per GPU, re-entrant per ctx */
typedef struct h2e_program h2e_program;  /* recorded shape of one workload */

const char* h2e_last_error(void);
const char* h2e_version(void);

/* replaces: Context::new + IntegerContext::new + RangeInfo::new (src/context.rs:136-143, :173-187) */
int h2e_ctx_create(int device, h2e_ctx** out);
void h2e_ctx_destroy(h2e_ctx* ctx);

/* ---- programs (shape recording; host only, no GPU needed) ------------------------------------- */
/* `emit_shape` = 0 records only the tape (values can be generated, no fixed/permutation artefacts). */

/* n independent `assign_w(a); assign_w(b); int_mul(a, b)` (IntegerChipOps::int_mul, src/circuit/integer_chip.rs:466-483).
 * inputs per instance: slots 2k, 2k+1 = a_k, b_k. */
int h2e_program_int_mul_batch(int field_pair, uint32_t n, int emit_shape, h2e_program** out);
/* body of test_integer_chip_st (src/tests/integer_chip.rs:11-55): add/sub/mul/div + div by zero.
 * inputs: a, b, c=a+b, d=a-b, e=a*b, f=a/b (6 slots). */
int h2e_program_integer_chip_st(int field_pair, int emit_shape, h2e_program** out);
/* body of test_native_ecc_chip_with_select_chip (src/tests/native_scalar_ecc_chip.rs:34-47) for one tile of
 * n points: assign_point x n, assign x n, msm_unsafe (EccChipScalarOps::msm_unsafe, src/circuit/ecc_chip.rs:373-408),
 * assign_point(expected), ecc_assert_equal.
 * inputs: 3n slots (x, y, z-flag) per point, n scalars, generator (x,y), r1 (x,y), r2 (x,y), expected (x,y,z). */
int h2e_program_msm_bn256_tile(uint32_t n_points, int emit_shape, h2e_program** out);
/* The same test body on a context without the select chip (NativeScalarEccContext::new_without_select_chip,
 * src/context.rs:190-207): msm_batch_on_group_non_zero_without_select_chip (src/circuit/ecc_chip.rs:91-221), groups of
 * two points, candidates chosen by bisec_candidate_non_zero (:913-933).  SURVEY.md §8(f)-3.  Same inputs. */
int h2e_program_msm_bn256_tile_no_select(uint32_t n_points, int emit_shape, h2e_program** out);
/* body of test_bls12_381_ecc_chip_over_bn256_fr (src/tests/general_scalar_ecc_chip.rs:14-49) for one tile of n points
 * (SURVEY.md 8(f)-2): GeneralScalarEccContext<bls12_381::G1Affine, bn256::Fr> - assign_point x n, scalar_integer_ctx.assign_w x n,
 * msm (general_scalar_ecc_chip.rs:93-168: scalars are 3-limb integers of the second integer context, decomposed limb by
 * limb into 324 one-bit windows), assign_point(expected), ecc_assert_equal.  Same input layout as h2e_program_msm_bn256_tile
 * with 6-word slots (points over bls12_381 Fq, scalars < bls12_381 r). */
int h2e_program_msm_bls12_381_tile(uint32_t n_points, int emit_shape, h2e_program** out);
/* check_pairing([(a, b), (-a, b)]) with G2 as constants (PairingChipOps::check_pairing,
 * src/circuit/pairing_chip.rs:173-176; shape of src/tests/native_scalar_pairing_chip.rs:67-97).
 * inputs: b.x.c0, b.x.c1, b.y.c0, b.y.c1, (-a).x, (-a).y, (-a).z, a.x, a.y, a.z. */
int h2e_program_pairing_check_bn256(int emit_shape, h2e_program** out);
/* check_pairing([(ac, b), (-a, bc)]) (shape of src/tests/general_scalar_pairing_chip.rs:74-105).
 * inputs: b.x.c0,b.x.c1,b.y.c0,b.y.c1, bc.x.c0,bc.x.c1,bc.y.c0,bc.y.c1, (-a).x,(-a).y,(-a).z, ac.x,ac.y,ac.z. */
int h2e_program_pairing_check_bls12_381(int emit_shape, h2e_program** out);
/* pairing(terms) with n_pairs (G1 assigned, G2 constant) pairs, and - with_expected - fq12_assert_eq against an Fq12
 * constant: the first block of the reference's pairing tests (PairingChipOps::pairing, src/circuit/pairing_chip.rs:157-171;
 * src/tests/native_scalar_pairing_chip.rs:20-65: 1 pair == native pairing(); general_scalar_pairing_chip.rs:20-72: the
 * product of 2 pairs == native).  curve: 0 = bn256, 1 = bls12_381.
 * inputs: per pair b.x.c0, b.x.c1, b.y.c0, b.y.c1; [expected: 12 W values c0.c0.c0, c0.c0.c1, c0.c1.c0, ... c1.c2.c1];
 * per pair a.x, a.y, a.z.  outputs (h2e_program_outputs): the 12 result integers' cells (limbs, native each). */
int h2e_program_pairing(int curve, uint32_t n_pairs, int with_expected, int emit_shape, h2e_program** out);
void h2e_program_destroy(h2e_program* p);

/* ---- shape artefacts: what Records holds besides advice values -------------------------------- */
typedef struct h2e_shape {
    int field_pair;
    uint32_t slot_words;        /* words per input slot (4 or 6) */
    uint32_t n_input_slots;
    uint64_t base_offset, range_offset, select_offset;   /* Context cursors after the run (src/context.rs:43-45) */
    uint64_t base_height, range_height, select_height;   /* Records heights (src/context.rs:297-299) */
    uint64_t base_rows, range_rows, select_rows;          /* rows to allocate per instance */
    uint64_t n_advice_cells;    /* assigned advice cells per instance (0 if emit_shape was 0) */
    uint64_t n_permutations;
    uint64_t n_dict;            /* fixed-value dictionary entries (entry 0 = None) */
    uint64_t n_fixed_patches;
    uint32_t n_segments;        /* engine launches per run */
    uint64_t n_ops;
    /* host arrays owned by the program (NULL when emit_shape was 0) */
    const uint64_t* dict;            /* [n_dict][4] canonical values */
    const uint32_t* base_fix;        /* [base_height][9] dictionary ids: coeff0..4, mul0, mul1, next, constant */
    const uint32_t* range_fix;       /* [range_height+1][2]: acc_lines, tag */
    const uint32_t* select_fix;      /* [select_height][2]: encode, is_lookup */
    const uint8_t* base_flags;       /* [base_height][5] bit0 = assigned, bit1 = permute (the bool of (Option<N>, bool)) */
    const uint8_t* range_flags;      /* [range_height+1][3] */
    const uint8_t* select_flags;     /* [select_height][2] */
    const uint32_t* permutations;    /* [n_permutations][2] cells: region<<30 | col<<27 | row (src/context.rs:300) */
    const uint32_t* fixed_patches;   /* [n_fixed_patches][4]: base row, fixed col, input slot, limb (-1 = value mod n) */
} h2e_shape;
int h2e_program_shape(const h2e_program* p, h2e_shape* out);

/* Result cells of the workload as cell references (region<<30 | col<<27 | row), program specific:
 * msm tile: res.x limbs, res.x native, res.y limbs, res.y native, res.z.  Returns the count. */
int h2e_program_outputs(const h2e_program* p, uint32_t* refs, uint32_t cap);
/* One entry of 8 words per engine launch of a run: n_strands, n_ops, advice cells written per instance
 * (0 unless emit_shape), per-strand Offset (base, range, select), n_params, first base row.  Returns the count. */
int h2e_program_launches(const h2e_program* p, uint64_t* out, uint32_t cap);
/* first row of the base / range / select array the k-th launch writes (a forked segment: of its strand 0): out[3] */
int h2e_program_launch_rows(const h2e_program* p, uint32_t launch, uint64_t* out);
/* diagnostics: opcodes (tape.h H2EOpcode) of the k-th launch's tape and the op indices its expansion's sub-ranges start at */
int h2e_program_tape_opcodes(const h2e_program* p, uint32_t launch, uint16_t* opcodes, uint32_t cap, uint32_t* subs, uint32_t subs_cap,
                             uint32_t* n_subs);
/* diagnostics: the order the packed expansion (batches smaller than half a wave) takes the k-th launch's sub-ranges in when a wave
 * holds 2 << groups_log2m1 of them: sub-range indices wave by wave, ~0u = empty slot.  Returns the entries; copies at most cap. */
int h2e_program_pack_order(const h2e_program* p, uint32_t launch, uint32_t groups_log2m1, uint32_t* out, uint32_t cap);
/* diagnostics: the k-th launch's value chain: out3 = {store ops per strand of a hint store, pieces of a compiled replay, field chain 0/1} */
int h2e_program_value_chain_kind(const h2e_program* p, uint32_t launch, uint32_t* out3);

/* ---- execution --------------------------------------------------------------------------------- */
/* Fill the advice values of n_instances instances.  d_base/d_range/d_select: batch-interleaved device arrays of
 * rows * cols * 4 * n_instances words (see Conventions); d_inputs as described above; d_status:
 * n_instances uint32 (or-ed, zero it first).  Asynchronous on `stream`: the engine fans the work out over
 * three internal streams of the context (expansion, early predictors / side segments, inverse fix-up), all of them
 * ordered after what is already queued on `stream`, and `stream` completes only when all of them have.
 * Re-entrancy: calls on one context are serialised on the host by the context's lock; every run owns its instance
 * table, workspace and events (a ring of H2E_STAT_PIPELINE_DEPTH job slots per context), so runs of the same or of
 * different programs may be queued back to back from any thread.  Different contexts share nothing. */
int h2e_run(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
            void* d_select, void* d_status, void* stream);

/* Pipelined submission for streaming jobs (a 2^20-point MSM is 1024 tiles through a ring of output buffers,
 * SURVEY.md 8d cfg 3).  Like h2e_run, but `stream` only carries the run's value chain and is NOT joined with the
 * engine's expansion streams: the value chain of the next h2e_submit overlaps this run's expansion.  *job identifies
 * the run; h2e_wait(job, s) makes stream `s` wait until every array of that run is complete.  At most
 * H2E_STAT_PIPELINE_DEPTH runs are in flight: a further submit first waits (on `stream`, not on the host) for the run
 * that used the same slot.  Runs in flight must use different advice arrays and status words.
 * RE-USE OF OUTPUT ARRAYS is ordered through `stream`: the run's kernels start behind the work queued on `stream` at the time of
 * the call (and behind the slot's previous run) and behind nothing else.  A caller that reads a finished run's arrays on ANOTHER
 * stream (export, digest, unit records) and then submits into the same arrays must make `stream` wait for those reads
 * (hipEventRecord on the reader's stream + hipStreamWaitEvent on `stream`) - or use one stream for both, as bench.py does. */
int h2e_submit(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
               void* d_select, void* d_status, void* stream, int* job);
int h2e_wait(h2e_ctx* ctx, int job, void* stream);

/* HALO2'S ADVICE COLUMNS STRAIGHT OUT OF THE EXPANSION (the reference's Records::_assign_to_*_chip, src/context.rs:
 * 310-541, without a pass of its own).  h2e_run whose full expansions store every assigned cell into per-instance column-major arrays
 * - d_cols_X = [n_instances][cols][rows][4 words], cols = 5 / 3 / 2, rows as h2e_program_shape reports, exactly what
 * h2e_export(H2E_LAYOUT_COLUMNS) of the run's arrays would give - while d_base / d_range / d_select stay the engine's working copy (the
 * cells ops read back are still written there; their contents after the run are NOT a witness).  The column arrays must be ZERO where the
 * shape leaves cells unassigned: the run writes assigned cells (and zeros around them inside the rows it passes) only; zero them once,
 * re-use them for every batch of the same program.  A wave's 64 lanes are 64 instances at one row and a column's cells reach memory as
 * 128-byte runs of one instance (four rows staged in LDS), the rate HBM takes write requests at.
 * n_instances: a multiple of 64; every program (each field pair has its column-emission unit); form: H2E_FORM_CANONICAL (H2E_FORM_MONTGOMERY:
 * h2e_export - Montgomery cells in the first pass would need 32-byte staging for the range array as well: 18 KB more LDS per wave, two
 * waves per compute unit instead of four). */
int h2e_run_columns(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range, void* d_select,
                    void* d_cols_base, void* d_cols_range, void* d_cols_select, int form, void* d_status, void* stream);

/* SEVERAL CALLER BATCHES AS ONE RUN.  n_batches (1 .. 16) batches of n_instances_each instances, every batch with its own inputs, its
 * own batch-interleaved arrays (over ITS n_instances_each instances) and its own status words - d_inputs[b], d_base[b], ... are host
 * arrays of n_batches device pointers, read before the call returns - executed as ONE run of n_batches x n_instances_each instances:
 * the same kernels and launches as a single batch of that size, every batch's cells in its own arrays (cell for cell what h2e_run of
 * that batch alone writes).  For a host with a STREAM of small batches - one GPU's share of a pairing job dealt over 8 GPUs is 2 (bls12_381)
 * or 8 (bn256) checks per step: a run's value chain costs ~2 ms of latency on one compute unit however small the batch, so eight 2-check
 * batches submitted together cost what one 16-check batch costs (0.07 instead of 0.20 ms per check, bench.py --group).  The consumer's
 * calls (h2e_export, h2e_digest, h2e_unit_records, h2e_check) take one batch's arrays at a time, as ever.  No stream digest. */
int h2e_run_batches(h2e_ctx* ctx, h2e_program* p, uint32_t n_batches, uint32_t n_instances_each, const void* const* d_inputs,
                    void* const* d_base, void* const* d_range, void* const* d_select, void* const* d_status, void* stream);
int h2e_submit_batches(h2e_ctx* ctx, h2e_program* p, uint32_t n_batches, uint32_t n_instances_each, const void* const* d_inputs,
                       void* const* d_base, void* const* d_range, void* const* d_select, void* const* d_status, void* stream, int* job);

/* MORE RUNS IN FLIGHT THAN ARRAY SETS FIT: h2e_ring.  A pipelined run holds its output arrays from its first kernel to its completion
 * and its value chain must be finished before its big expansion can stream; with two array sets (2 x 110 GB for 64 x 1024-point MSM
 * tiles) the step is half a run's latency, and a third set does not fit 288 GB.  But ONE launch of such a program owns most rows (the MSM's
 * window strands: 81-89 % of every array) and nothing writes them before that launch's own value chain stores its operand cells, late in
 * the run.  A ring backs the rows of the program's biggest launch by TWO physical copies and all other rows by `depth`, mapped (HIP
 * virtual-memory API) into lcm(2, depth) virtual array sets: run k uses set k - rest copy k mod depth, big-launch copy k mod 2 - and
 * whatever of run k writes the big launch's rows waits for the completion of run k - 2.  depth 3: three runs in flight in 2.2 sets.
 *   - set H2E_OPT_PIPELINE_DEPTH = depth first; submit the ring's runs in order (k = 0, 1, 2, ...) and nothing else on the context meanwhile;
 *   - h2e_ring_arrays(k) are run k's arrays for the consumer's calls (h2e_export, h2e_digest, h2e_unit_records, h2e_check): ordinary
 *     batch-interleaved arrays.  Run k + 2 writes the rows of the big launch (h2e_ring_info: launch index; h2e_program_launch_rows) of
 *     the same physical memory; all other rows are run k + depth's.  What orders the two: by default run k + 2's writers of those rows
 *     wait for run k's COMPLETION - enough for a consumer that takes the shared rows through the stream digest (h2e_ring_submit_digest)
 *     or not at all (status words, unit records: the result cells lie outside the shared launch).  A consumer that READS them after
 *     the run (export, h2e_digest, h2e_check) says where its reads end: h2e_ring_release(ring, k, consumer_stream), called before
 *     h2e_ring_submit(k + 2) - run k + 2's writers of the shared rows, and run k + depth as a whole, then wait for that point of the
 *     consumer's stream.  Submit on ANOTHER stream than the
 *     consumer's (a run starts behind what its submission stream holds: on the consumer's stream it would start behind run k's
 *     completion and two runs would be in flight, not three);
 *   - h2e_wait(job) as for h2e_submit.  Errors: H2E_ERR_HIP if the device cannot map the memory (no fallback: use h2e_submit). */
typedef struct h2e_ring h2e_ring;
int h2e_ring_create(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, uint32_t depth, h2e_ring** out);
void h2e_ring_destroy(h2e_ring* ring);
int h2e_ring_arrays(const h2e_ring* ring, uint64_t k, void** d_base, void** d_range, void** d_select);
/* out[0..2] = bytes of a full array set (base, range, select), [3..5] = of each, the bytes the two shared copies back, [6] = physical
 * bytes of the ring, [7] = the shared launch (index in h2e_program_launches), [8] = virtual array sets, [9] = depth.  Returns 10. */
int h2e_ring_info(const h2e_ring* ring, uint64_t* out, uint32_t cap);
int h2e_ring_release(h2e_ring* ring, uint64_t k, void* stream);
int h2e_ring_submit(h2e_ring* ring, uint64_t k, const void* d_inputs, void* d_status, void* stream, int* job);
int h2e_ring_submit_digest(h2e_ring* ring, uint64_t k, const void* d_inputs, void* d_status, void* d_digests, void* stream, int* job);

/* Options / statistics of a context.  Tuning knobs are read from the environment once, at h2e_ctx_create
 * (H2E_X_SPLIT, H2E_X_SPLIT_MIN_LANES, H2E_X_PARTS); nothing reads the environment while a run is queued.
 * PROCESS-wide, not per context (every context of a process sees the same value; set them before the first context runs):
 * the kernel-side fields of the experiment knob H2E_TUNE and the test hook H2E_OPT_TEST_SCAN_FALLBACK / the counter
 * H2E_STAT_SCAN_FALLBACKS.  Everything else - streams, job slots, workspaces, options, statistics, the op-program cache -
 * belongs to its context: two contexts on one device share nothing (tests/test_threads_gpu.py). */
#define H2E_OPT_X_SPLIT_PCT 1          /* a big expansion goes out as several launches (three; H2E_X_PARTS): percent of sub-ranges in the first (0 = off) */
#define H2E_OPT_X_SPLIT_MIN_LANES 2    /* ... if it has at least this many lanes */
#define H2E_OPT_TEST_SKIP_EXPANSION 3  /* TEST HOOK: leave out the full expansion of cut segment <value> (-1: of every cut
                                          segment but the last; INT64_MIN: off).  Rows are missing from such a run: every
                                          status word gets H2E_ST_TEST_HOOK. */
#define H2E_OPT_PIPELINE_DEPTH 4       /* job slots in use = runs h2e_submit keeps in flight (1 .. H2E_STAT_MAX_PIPELINE_DEPTH = 32, default 2).
                                          Every slot has its own workspace and streams; call with no run in flight.  2 for the 64-tile
                                          MSM (a buffer set is 110 GB), 4 for a full 64-check bn256 batch, 16 for smaller pairing
                                          batches (a check's value chain is ~2 ms of latency on one compute unit; such a run lives in one
                                          stream).  A slot's FIRST run allocates its workspace and creates its streams (~10 ms on the
                                          host); the process wants GPU_MAX_HW_QUEUES >= depth + 12 set before HIP initialises, and more
                                          than ~24 streams in use are time-sliced by the hardware. */
#define H2E_OPT_TEST_SCAN_FALLBACK 5    /* TEST HOOK: bit mask - the MSM scan predictors treat some of their (valid) start values as degenerate
                                          and walk the real chain instead (1 window chunks, 2 tail chunk sums, 4 tail in-chunk starts).
                                          The results are the same; what it covers is the fallback path.  Process-wide; 0 = off. */
#define H2E_OPT_PREFAULT_HBM 6          /* one throw-away allocate / fill / free of <value> percent of the device's free memory, now
                                          (synchronous, ~0.3 s for 270 GB).  The first process that streams into HBM nobody has
                                          written since the device booted runs at about half the rate of any later one; a host
                                          that cares about its first runs calls this once after h2e_ctx_create. */
#define H2E_OPT_OP_CACHE_CAP 7           /* operator API: programs the context keeps for ops it has seen (default 4096, >= 1).  Beyond it the
                                          least recently used programs that no call is running are freed with their device tapes. */
int h2e_ctx_set_option(h2e_ctx* ctx, int option, int64_t value);
#define H2E_STAT_LAST_SPLIT_SEGMENTS 1 /* segments of the last run whose expansion went out as several launches */
#define H2E_STAT_RUNS 2
#define H2E_STAT_PIPELINE_DEPTH 3
#define H2E_STAT_MAX_PIPELINE_DEPTH 4
#define H2E_STAT_SCAN_FALLBACKS 5      /* lanes of the MSM scan predictors that had to walk the real chain so far (process-wide; synchronises) */
#define H2E_STAT_OP_CACHE_HITS 6       /* operator API: ops whose program came from the context's cache (keyed by op, arguments, */
#define H2E_STAT_OP_CACHE_MISSES 7     /* operand handles, cursors, heights, msm prefix) / ops that had to be recorded */
#define H2E_STAT_OP_CACHE_EVICTIONS 8  /* programs the cache let go (H2E_OPT_OP_CACHE_CAP) */
#define H2E_STAT_OP_CACHE_SIZE 9
#define H2E_STAT_HW_QUEUES 10          /* hardware queues the process's streams are mapped onto (GPU_MAX_HW_QUEUES as the environment had it when HIP
                                          initialised; default 4) */
#define H2E_STAT_HW_QUEUES_WANTED 11   /* ... and what the context's pipeline depth wants (depth + 12; 1 without pipelining).  h2e_ctx_set_option(
                                          H2E_OPT_PIPELINE_DEPTH) succeeds either way, but leaves a text in h2e_last_warning() and one line on stderr
                                          when the process has fewer: streams that share a queue serialise */
int64_t h2e_ctx_get_stat(h2e_ctx* ctx, int stat);
/* text of the last call's warning on this thread ("" if it had none): a call that succeeded but will not perform as asked */
const char* h2e_last_warning(void);

/* Named entry points of SURVEY.md §8(b): build-or-reuse the program for the shape, then run it. */
int h2e_int_mul_batch(h2e_ctx* ctx, int field_pair, uint32_t n, uint32_t n_instances, const void* d_inputs, void* d_base,
                      void* d_range, void* d_select, void* d_status, void* stream);
int h2e_msm_bn256_tile(h2e_ctx* ctx, uint32_t n_points, uint32_t n_tiles, const void* d_inputs, void* d_base, void* d_range,
                       void* d_select, void* d_status, void* stream);
int h2e_pairing_check_bn256(h2e_ctx* ctx, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
                            void* d_select, void* d_status, void* stream);
int h2e_pairing_check_bls12_381(h2e_ctx* ctx, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
                                void* d_select, void* d_status, void* stream);

/* ---- hand-off to the consumer (device half) ------------------------------------
 * halo2 keeps one array per advice column and its field elements are Montgomery-form [u64; 4]; the reference's
 * Records::_assign_to_{base,range,select}_chip (src/context.rs:310-541) copy row-major `[(Option<N>, bool); COLS]`
 * cells into them one by one, and every value crosses utils.rs:10-17 (`bn_to_field`).  h2e_export turns the
 * batch-interleaved array of one region (0 base, 1 range, 2 select) of a run into one array per instance,
 *   H2E_LAYOUT_ROWS     [instance][row][COLS][4 words]   (the reference's Records layout), or
 *   H2E_LAYOUT_COLUMNS  [instance][COLS][row][4 words]   (halo2's advice columns),
 * with cells the shape leaves unassigned as zero (programs recorded with emit_shape = 0 have no flags: their cells are
 * copied as they are), as canonical little-endian values (H2E_FORM_CANONICAL, what field_to_bn sees, utils.rs:4-8) or
 * as Montgomery-form words (H2E_FORM_MONTGOMERY: x * 2^256 mod n, the in-memory form of halo2's Fr).
 * Asynchronous on `stream`. */
#define H2E_LAYOUT_ROWS 0
#define H2E_LAYOUT_COLUMNS 1
#define H2E_FORM_CANONICAL 0
#define H2E_FORM_MONTGOMERY 1
int h2e_export(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, int region, int layout, int form, const void* d_batch,
               void* d_out, void* stream);

/* ---- shape artefacts in the prover's layout, made on the device -----------------
 * What Records holds besides advice values is shape-only (h2e_program_shape keeps it as dictionary ids / flag bytes / cell
 * pairs on the host); these three put it into HBM in the form halo2 consumes:
 *  h2e_export_fixed: the fixed cells of one region (base 9, range 2, select 2 columns: src/context.rs:367-375,
 *    src/circuit/range_chip.rs:88-92, select_chip.rs:48-52) as [instance][COLS][row][4 words] (H2E_LAYOUT_COLUMNS) or
 *    [instance][row][COLS][4] (H2E_LAYOUT_ROWS), None -> 0, canonical or Montgomery form.  Fixed cells are the same for every
 *    instance except those made from instance inputs (the G2 constants of a pairing check): pass the run's d_inputs for them.
 *  h2e_range_table: the 18-bit tagged range lookup table of RangeChip::init_table (src/circuit/range_chip.rs:230-258) -
 *    row (tag, value) for tag in 0..=18, value in 0..2^tag - as [2][H2E_RANGE_TABLE_ROWS][4 words] (tag column, value column).
 *  h2e_export_copy_constraints: the permutation list (src/context.rs:300, _assign_permutation :523-541) as
 *    [n_permutations][4] uint32 = (advice column a, row a, advice column b, row b), advice columns numbered 0-4 base,
 *    5-7 range, 8-9 select. */
#define H2E_RANGE_TABLE_ROWS 524287u
int h2e_export_fixed(h2e_ctx* ctx, h2e_program* p, int region, int layout, int form, uint32_t n_instances, const void* d_inputs, void* d_out,
                     void* stream);
int h2e_range_table(h2e_ctx* ctx, int form, void* d_out, void* stream);
int h2e_export_copy_constraints(h2e_ctx* ctx, h2e_program* p, void* d_out, void* stream);

/* ---- the reference's acceptance criterion, on the device, for every instance of a run -----------------------------------
 * Every test of the reference ends in `MockProver::run(k, &circuit, vec![]).verify() == Ok(())` (src/tests/mod.rs:117-132):
 * the arrays satisfy the base gate (src/circuit/base_chip.rs:50-69), the three range accumulation gates and the two range
 * lookups (src/circuit/range_chip.rs:119-220, table :230-258), the select chip's lookup_any (src/circuit/select_chip.rs:71-88)
 * and every copy constraint (src/context.rs:523-541).  h2e_check evaluates exactly those over the batch-interleaved advice
 * arrays a run of a PROGRAM left in HBM and the program's own fixed cells, flags and permutation list (unassigned / unset
 * cells = 0, as in MockProver), for all n_instances at once; `p` must have been recorded with its shape (emit_shape = 1) and
 * d_inputs is the run's input vector (fixed cells made from instance inputs - the G2 constants - are recomputed from it by
 * the checker's own arithmetic).  (An operator-API context has no program handle to pass here: a host that wants its
 * sequence of ops checked records it as a program.)
 * d_fail = uint64 [n_instances][2 * H2E_CHECK_CLASSES]: per instance the number of failing rows (pairs) of each class, then
 * the lowest failing row (pair index) of each class, ~0 when none.  An instance passes iff its first five words are zero.
 * `classes` = bit mask of H2E_CHECK_* to evaluate (0 = all).  Asynchronous on `stream`. */
#define H2E_CHECK_BASE_GATE 0
#define H2E_CHECK_RANGE_GATE 1
#define H2E_CHECK_RANGE_LOOKUP 2
#define H2E_CHECK_SELECT_LOOKUP 3
#define H2E_CHECK_COPY 4
#define H2E_CHECK_CLASSES 5
int h2e_check(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, const void* d_base, const void* d_range,
              const void* d_select, uint32_t classes, void* d_fail, void* stream);

/* ---- operator API: a device-resident Context -----------------------------------------------------
 * The reference's operator surface is a Context you call chip ops on (`IntegerChipOps` src/circuit/integer_chip.rs:15-70,
 * `EccChipBaseOps` / `EccChipScalarOps` src/circuit/ecc_chip.rs:79-430, `PairingChipOps` src/circuit/pairing_chip.rs:157-176); its own
 * seam for doing part of the work elsewhere is fork-at-offset / merge (`ParallelClone`, ecc_chip.rs:64-77).  h2e_records is that
 * Context for a batch of n_instances instances which run the same ops on different values: the advice arrays live in HBM
 * (batch-interleaved, `rows` = the capacity, like HALO2ECC_S_MAX_ROWS of src/context.rs:36), cursors, heights, the msm prefix
 * (`NativeScalarEccContext.1`, native_scalar_ecc_chip.rs:173-178) and the shape artefacts on the host.  Every op appends its rows
 * at the current offsets - exactly the rows the reference's op writes when called at that point of a Context - takes its operands
 * as handles (cell references to rows written earlier, plus the static `times`), and returns handles.  Values never cross the
 * boundary except as inputs: d_inputs = device [n_instances][slots of the op][slot_words] canonical words.
 * field_pair = the W field of the integer chip / the base field of the curve; scalar_field = -1 for a NativeScalarEccContext,
 * H2E_FIELD_BLS12_381_FR for GeneralScalarEccContext<bls12_381::G1Affine, bn256::Fr> (scalars are h2e_int of that field then;
 * native scalars use h2e_int.native only).  Ops are asynchronous on `stream` like h2e_run; status words accumulate (or). */
typedef struct h2e_records h2e_records;
typedef struct h2e_int { uint32_t limbs[4]; uint32_t native; uint32_t times; } h2e_int;      /* AssignedInteger (src/assign.rs:31-37) */
typedef struct h2e_point { h2e_int x, y; uint32_t z; } h2e_point;                            /* AssignedPoint (src/assign.rs:46-51) */
typedef struct h2e_g2 { h2e_int x0, x1, y0, y1; uint32_t z; } h2e_g2;                         /* AssignedG2Affine (src/assign.rs:171-192) */
#define H2E_RECORDS_EMIT_SHAPE 1       /* emit_shape / flags: keep the fixed cells, flags and permutation list of every op (h2e_records_shape) */
#define H2E_RECORDS_NO_SELECT_CHIP 2   /* NativeScalarEccContext::new_without_select_chip (src/context.rs:201-205): msm prefix usize::MAX,
                                          msm_unsafe / ecc_mul take the bisection form (src/circuit/ecc_chip.rs:91-221, dispatch :373-408);
                                          with h2e_records_attach: msm_prefix0 = UINT64_MAX */
int h2e_records_create(h2e_ctx* ctx, int field_pair, int scalar_field, uint32_t n_instances, uint64_t base_rows, uint64_t range_rows,
                       uint64_t select_rows, int emit_shape /* H2E_RECORDS_* flags; 0 / 1 as before */, h2e_records** out);
/* The splice seam itself (ParallelClone: clone_with_offset + merge + apply_offset_diff, src/circuit/ecc_chip.rs:64-77, used at
 * :289-352; NativeScalarEccContext clone / merge, src/circuit/native_scalar_ecc_chip.rs:50-90, msm prefix :173-178): a records
 * object over arrays the CALLER allocated (batch-interleaved, capacity_rows[3] rows), whose ops start at the caller's cursors
 * offset0 = (base, range, select) and msm prefix - i.e. a forked context that writes its disjoint rows into the caller's Records.
 * Rows below the offsets are the caller's and are never touched; heights start at the offsets; h2e_records_shape then reports
 * the new offsets / heights (what apply_offset_diff / merge need) and the fixed cells / flags / permutations of the spliced rows.
 * The engine never frees the arrays.  d_status: n_instances uint32 (or-ed). */
int h2e_records_attach(h2e_ctx* ctx, int field_pair, int scalar_field, uint32_t n_instances, void* d_base, void* d_range, void* d_select,
                       void* d_status, const uint64_t capacity_rows[3], const uint64_t offset0[3], uint64_t msm_prefix0, int emit_shape,
                       h2e_records** out);
void h2e_records_destroy(h2e_records* rec);
int h2e_records_arrays(h2e_records* rec, void** d_base, void** d_range, void** d_select, void** d_status);
/* offsets, heights, accumulated fixed cells / flags / permutations over rows [0, capacity); *_rows = the capacity;
 * fixed_patches[k] = [row, fixed col, op index << 16 | input slot, limb] */
int h2e_records_shape(const h2e_records* rec, h2e_shape* out);
int h2e_op_assign_w(h2e_records* rec, const void* d_inputs /* 1 slot */, h2e_int* out, void* stream);
int h2e_op_assign(h2e_records* rec, const void* d_inputs /* 1 slot */, uint32_t* out_cell, void* stream);
#define H2E_INT_ADD 0
#define H2E_INT_SUB 1
#define H2E_INT_MUL 2
#define H2E_INT_DIV 3      /* out_cond = the is_b_zero condition cell */
#define H2E_INT_REDUCE 4   /* b unused */
/* the rest of IntegerChipOps (src/circuit/integer_chip.rs:15-70) through the same entry: */
#define H2E_INT_NEG 5            /* int_neg :439-464; b unused */
#define H2E_INT_SQUARE 6         /* int_square :614-616; b unused */
#define H2E_INT_UNSAFE_INVERT 7  /* int_unsafe_invert :485-491; b unused */
#define H2E_INT_IS_ZERO 8        /* is_int_zero :540-578; out unused, out_cond = the condition cell */
#define H2E_INT_IS_EQUAL 9       /* is_int_equal :47-54; out unused, out_cond = the condition cell */
#define H2E_INT_ASSERT_EQUAL 10  /* assert_int_equal :600-612; out, out_cond unused */
int h2e_op_int(h2e_records* rec, int which, const h2e_int* a, const h2e_int* b, h2e_int* out, uint32_t* out_cond, void* stream);
int h2e_op_int_mul_small_constant(h2e_records* rec, const h2e_int* a, uint64_t k, h2e_int* out, void* stream);         /* :618-658 */
int h2e_op_assign_int_constant(h2e_records* rec, const uint64_t* w_words /* canonical, slot_words */, h2e_int* out, void* stream);   /* :580-598 */
int h2e_op_bisec_int(h2e_records* rec, uint32_t cond_cell, const h2e_int* a, const h2e_int* b, h2e_int* out, void* stream);       /* :660-681 */
/* Fq2 / Fq6 / Fq12 ops on assigned elements (Fq2ChipOps / Fq6ChipOps / Fq12ChipOps, src/circuit/fq12.rs:24-459): what a circuit
 * calls between pairings.  An element of degree k is k h2e_int in the order of fq12_assign_constant (fq12.rs:453-458):
 * c0.c0.c0, c0.c0.c1, c0.c1.c0, ...  `a`, `b`, `out`: arrays of `degree` integers (b / out NULL where the op has none). */
#define H2E_FQ_ADD 0
#define H2E_FQ_SUB 1
#define H2E_FQ_MUL 2
#define H2E_FQ_SQUARE 3
#define H2E_FQ_NEG 4
#define H2E_FQ_DOUBLE 5
#define H2E_FQ_CONJUGATE 6            /* degree 2, 12 */
#define H2E_FQ_UNSAFE_INVERT 7
#define H2E_FQ_MUL_BY_NONRESIDUE 8    /* degree 2, 6 */
#define H2E_FQ_FROBENIUS_MAP 9        /* imm = power */
#define H2E_FQ_CYCLOTOMIC_SQUARE 10   /* degree 12 */
#define H2E_FQ_REDUCE 11
#define H2E_FQ_ASSERT_EQUAL 12        /* no out */
int h2e_op_fq(h2e_records* rec, int degree, int which, const h2e_int* a, const h2e_int* b, uint64_t imm, h2e_int* out, void* stream);
int h2e_op_assign_points(h2e_records* rec, uint32_t n, const void* d_inputs /* (x, y, z) x n */, h2e_point* out, void* stream);
int h2e_op_assign_scalars(h2e_records* rec, uint32_t n, const void* d_inputs /* n slots */, h2e_int* out, void* stream);
/* EccChipScalarOps::msm_unsafe on assigned points / scalars (ecc_chip.rs:373-408).  d_inputs: generator x, y, then the blinding
 * points r1 (x, y), r2 (x, y) the reference draws inside (quirk Q1).  A failing instance reports H2E_ST_RETRY_* like UnsafeError. */
int h2e_op_msm_unsafe(h2e_records* rec, uint32_t n, const h2e_point* points, const h2e_int* scalars, const void* d_inputs, h2e_point* out,
                      void* stream);
int h2e_op_ecc_assert_equal(h2e_records* rec, const h2e_point* a, const h2e_point* b, void* stream);
/* the complete-addition / curvature surface of EccChipBaseOps (src/circuit/ecc_chip.rs:441-812): */
typedef struct h2e_point_c { h2e_point p; h2e_int cv; uint32_t cz; } h2e_point_c;              /* AssignedPointWithCurvature (src/assign.rs:59-65) */
int h2e_op_to_point_with_curvature(h2e_records* rec, const h2e_point* a, h2e_point_c* out, void* stream);        /* :695-708 */
int h2e_op_ecc_reduce_with_curvature(h2e_records* rec, const h2e_point* a, h2e_point_c* out, void* stream);      /* :677-693 (ecc_reduce, assign_identity) */
int h2e_op_ecc_double(h2e_records* rec, const h2e_point_c* a, h2e_point* out, void* stream);                      /* :630-642 */
int h2e_op_ecc_add(h2e_records* rec, const h2e_point_c* a, const h2e_point* b, h2e_point* out, void* stream);     /* :606-628 */
int h2e_op_ecc_neg(h2e_records* rec, const h2e_point* a, h2e_point* out, void* stream);                           /* :660-666 */
int h2e_op_ecc_encode(h2e_records* rec, const h2e_point* a, uint32_t* out_cells3, void* stream);                  /* :710-732 */
int h2e_op_ecc_mul(h2e_records* rec, const h2e_point* a, const h2e_int* scalar, const void* d_inputs /* as msm_unsafe */, h2e_point* out,
                   void* stream);                                                                                 /* :418-420 */
int h2e_op_assign_constant_point(h2e_records* rec, const uint64_t* x_words, const uint64_t* y_words, int is_identity, h2e_point* out,
                                 void* stream);                                                                   /* :441-456 */
int h2e_op_bisec_point_with_curvature(h2e_records* rec, uint32_t cond_cell, const h2e_point_c* a, const h2e_point_c* b, h2e_point_c* out,
                                      void* stream);                                                              /* :562-578 */
int h2e_op_assign_cache_point(h2e_records* rec, const h2e_point_c* p, uint64_t group, uint64_t selector, void* stream);   /* :779-788 */
/* :790-812; the reference is handed the chosen candidate, here it is picked on the device by the value of the index cell */
int h2e_op_assign_selected_point(h2e_records* rec, uint32_t n, const h2e_point_c* candidates, uint32_t index_cell, uint64_t group,
                                 h2e_point_c* out, void* stream);
int h2e_op_assign_g2_constant(h2e_records* rec, const void* d_inputs /* x.c0, x.c1, y.c0, y.c1 */, h2e_g2* out, void* stream);
int h2e_op_check_pairing(h2e_records* rec, uint32_t n_pairs, const h2e_point* g1, const h2e_g2* g2, void* stream);
/* PairingChipOps::pairing on assigned terms (src/circuit/pairing_chip.rs:157-171): out12 = the 12 integers of the Fq12 result */
int h2e_op_pairing(h2e_records* rec, uint32_t n_pairs, const h2e_point* g1, const h2e_g2* g2, h2e_int* out12, void* stream);

/* On-device consumer for streaming jobs (SURVEY.md 8d cfg 3, 8e): a 32-byte digest per instance of one region's
 * batch-interleaved array, d_digests = [n_instances][4] words:
 *   digest[j] = sum over assigned cells (row, col) of sm(w_j ^ sm(row * COLS + col) ^ j * 0xA24BAED4963EE407)  mod 2^64,
 * sm = the SplitMix64 finaliser (z += 0x9E3779B97F4A7C15; z = (z ^ z >> 30) * 0xBF58476D1CE4E5B9;
 * z = (z ^ z >> 27) * 0x94D049BB133111EB; z ^ z >> 31), w_0..w_3 the cell's canonical words.  Programs recorded
 * without their shape digest every cell of the array.  Asynchronous on `stream`. */
int h2e_digest(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, int region, const void* d_batch, void* d_digests, void* stream);

/* The per-unit record table of a job's ONE collective (units = MSM tiles / pairing instances, each its own
 * `Context` in reference terms, src/context.rs:136-143).  One kernel on `stream` writes, for every instance u of a finished run,
 * h2e_unit_record_words(p) 64-bit words to d_out + u * out_stride_words:
 *   [0]      the unit's status word (as int32, sign-extended)
 *   [1..3]   the program's Offset: base / range / select rows consumed (src/circuit/ecc_chip.rs:36-62)
 *   [4..]    the result point as its cells' values: x limbs, then y limbs (2 words each - a limb is < 2^128), then z (1 word);
 *            zero for programs without a result point (the pairing checks).  3-limb curves: 13 words, bls12_381: 17
 *   [last 12] d_digests[region][instance][4] (what h2e_run_digest / h2e_submit_digest left), zero if d_digests is NULL
 * = 29 words (bn256 workloads) or 33 (bls12_381 tiles).  out_stride_words >= the record size lets the caller keep columns of
 * its own in the same table (e.g. a leading global unit index, so that the table is what it hands to ncclAllGather as it is:
 * INTEGRATION.md "The gather").  Reads d_base (batch-interleaved base array of the run) and d_status only.
 * A program without outputs (no result point) gets the 3-limb record size - 29 words, point words zero - by convention; a program
 * whose outputs are not a point (h2e_program_pairing: an Fq12), or whose output references are not absolute base-array cells inside
 * the program's rows, is refused (H2E_ERR_INVALID). */
int h2e_unit_record_words(const h2e_program* p);
int h2e_unit_records(h2e_ctx* ctx, const h2e_program* p, uint32_t n_instances, const void* d_base, const void* d_status,
                     const void* d_digests /* or NULL */, void* d_out, uint32_t out_stride_words, void* stream);

/* The same consumer at no extra pass: the STREAM DIGEST.  h2e_run_digest / h2e_submit_digest are h2e_run / h2e_submit whose
 * expansion (and inverse fix-up) kernels add every cell they store to a position-keyed linear checksum while the value is
 * still in registers - a streaming job that only needs a fingerprint of each tile (SURVEY.md 8d cfg 3, 8e) does not read its
 * 78 GB of cells a second time (h2e_digest above is such a second pass: it more than doubles the step).
 * d_digests = [3][n_instances][4] words (region-major: base, range, select), zeroed by the engine, complete with the run:
 *   pos = row * COLS + col (32 bit), h = pos * 0x9E3779B1 (mod 2^32), k0 = (h ^ (h >> 15)) | 1, k1 = (h * 0x85EBCA77 + 0xC2B2AE3D) | 1
 *   digest[j] = sum over assigned cells of lo32(w_j) * k0 + hi32(w_j) * k1        (mod 2^64), w_0..w_3 the cell's canonical words.
 * Linear in the values, keyed by position: any single-cell change and any swap of two different cells changes it; it is a
 * transport checksum, not a cryptographic hash.  tests: == the oracle's stream digest of its Records (every cell, at full size). */
int h2e_run_digest(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
                   void* d_select, void* d_status, void* d_digests, void* stream);
int h2e_submit_digest(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
                      void* d_select, void* d_status, void* d_digests, void* stream, int* job);

/* Timing hook used by bench.py: HIP events recorded by the engine on the stream each kernel group is launched
 * on.  Returns the number of launched segments and fills two numbers per segment: ms[2i] = value chain
 * (predictor kernels + values-only replay), ms[2i+1] = full expansion (the inverse fix-up runs on its own stream and is not included; call after synchronising).
 * Refers to the most recently queued run; returns 0 when that run was queued with profiling off. */
int h2e_last_run_launch_ms(h2e_ctx* ctx, float* ms, uint32_t cap);
int h2e_set_profiling(h2e_ctx* ctx, int enable);
/* the same for the last run queued on job slot `job` (h2e_submit); waits on the host for that run to complete */
int h2e_job_launch_ms(h2e_ctx* ctx, int job, float* ms, uint32_t cap);
/* Companion of h2e_last_run_launch_ms: counts[i] = kernel launches the full expansion of segment i went out as in the
 * last run (1, or 2 when a big expansion was split - ms[2i+1] then brackets both; see h2e_capi.cpp `expand`). */
int h2e_last_run_expansion_launches(h2e_ctx* ctx, uint32_t* counts, uint32_t cap);

/* ---- test hook ----------------------------------------------------------------------------------
 * The digit-row primitives of the pairings' value chain (csrc/engine.hip DigitRow: one 16-lane row per number, one lane per
 * 32-bit digit) on caller-chosen operands - the patterns random data never produces (a carry through a run of 0xffffffff digits,
 * quotient estimates at exact multiples of w): tests/test_digit_rows_gpu.py.  op 0: normalize(columns lo, hi), 2: Montgomery
 * product, else: reduce_columns; d_in = [n_cases][2][16] u32, d_out = [n_cases][16] u32; field_pair 0 or 1, whose constants a
 * run of any program of that pair has put on the device. */
int h2e_selftest_digit_rows(int field_pair, uint32_t op, uint32_t n_cases, const void* d_in, void* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* H2E_H */
