// gfx950 witness engine: replays a witness tape (tape.h) for many instances / strands at once.
//
// Mapping: one lane per (strand, instance), instance minor.  All lanes of a launch execute the same op at the
// same time (the tape is wave-uniform, fetched through the scalar cache), so there is no divergence except inside
// the two modular inversions.  Advice arrays are *batch-interleaved*: [row][col][half][instance][2 x u64] - the rows
// and columns are exactly those the reference's (forked) context would have written (src/context.rs:610-632,
// 803-815, 722-735), a cell is two 16-byte halves (canonical little-endian bn256-Fr), and the instances of the batch
// are the minor dimension.  The 64 lanes of a wave are 64 instances at the same (row, col): every store instruction
// of a wave is one contiguous 1 KB run, unassigned cells are never touched (no bytes beyond the 32 B per assigned
// cell), and a wave walks through memory sequentially row after row.  With one instance the layout is the
// reference's row-major `[row][col][4 x u64]`.  Operands are read back from those arrays through cell references.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "tape.h"
#include "wide_int.h"

// ------------------------------------------------------------------------------------------------
// Translation units.  The build compiles this file once per field pair (-DH2E_FP_ONLY=0|1|2, side by side: the
// kernels of one pair are ~1.5 minutes of compile time, all three in one unit 5): unit k holds the templated kernels
// of pair k behind h2e_engine_launch_fpK / h2e_engine_predict_fpK / ..., unit 0 also the kernels that do not depend on
// the pair and the dispatchers under the plain names the C-ABI layer calls.  Without H2E_FP_ONLY everything is one unit.
#ifndef H2E_FP_ONLY
#define H2E_FP_ONLY -1
#endif
#define H2E_HAS_FP(id) (H2E_FP_ONLY < 0 || H2E_FP_ONLY == (id))
// The COLUMN-EMISSION unit (-DH2E_COLS, with H2E_FP_ONLY = k): this file once more for pair k, holding ONE kernel - the full expansion
// that stores halo2's per-instance advice columns itself (h2e_run_tape_cols, behind h2e_engine_launch_cols_fpK) - with the emission layer
// below compiled in its staging form.  A unit of its own, so that the plain expansion's register allocation (249 of 256) sees none of it.
#ifdef H2E_COLS
#define H2E_COLS_ON 1
#else
#define H2E_COLS_ON 0
#endif
#define H2E_COMMON_UNIT (H2E_FP_ONLY <= 0 && !H2E_COLS_ON)
#define H2E_CAT2(a, b) a##b
#define H2E_CAT(a, b) H2E_CAT2(a, b)
#if H2E_COLS_ON
#define H2E_UNIT(name) H2E_CAT(name##_colsfp, H2E_FP_ONLY)
#elif H2E_FP_ONLY >= 0
#define H2E_UNIT(name) H2E_CAT(name##_fp, H2E_FP_ONLY)
#else
#define H2E_UNIT(name) name
#endif
// symbols with external linkage that every unit has its own copy of
#define g_fc H2E_UNIT(g_fc)
#define h2e_fixup_inverses H2E_UNIT(h2e_fixup_inverses)

// ------------------------------------------------------------------------------------------------
// field-pair traits (compile-time sizes; values come from H2EFieldConsts)
struct FP_BN256_FQ {   // bn256 Fq over bn256 Fr
    static constexpr int ID = 0;
    static constexpr int L = 3, WW = 4, K = 254, CEIL = 254, MC = 3, RC = 1, PW = 1;
    static constexpr int NSUB = 1;   // a canonical W value is < 2n: "mod n" is one conditional subtraction
};
struct FP_BLS_FQ {     // bls12_381 Fq over bn256 Fr
    static constexpr int ID = 1;
    static constexpr int L = 4, WW = 6, K = 381, CEIL = 381, MC = 5, RC = 2, PW = 2;
    static constexpr int NSUB = 0;   // needs a real reduction
};
struct FP_BLS_FR {     // bls12_381 Fr over bn256 Fr
    static constexpr int ID = 2;
    static constexpr int L = 3, WW = 4, K = 255, CEIL = 255, MC = 3, RC = 1, PW = 1;
    static constexpr int NSUB = 2;   // bls12_381 r < 3n
};
template <class FP>
struct FPX {
    static constexpr int S = 2 * FP::CEIL + 12;           // a*b < 2^S
    static constexpr int XW = (S + 63) / 64;
    static constexpr int QW = (S - FP::K + 1 + 63) / 64;  // words of a quotient d
    static constexpr int AW = (FP::CEIL + 6 + 63) / 64;   // words of a composed operand (< 2^(CEIL+6))
    static constexpr int AL = (FP::CEIL + 6 + 31) / 32;   // ... and its significant 32-bit limbs
};
static constexpr int NK = 254;  // bit length of bn256 Fr modulus

// Field-pair constants live in constant memory: every access has a wave-uniform address, so the compiler emits
// scalar loads (s_load through the scalar cache into SGPRs) instead of per-lane global loads.  With a plain
// device pointer each modulus / Barrett / ceil-table word was a dependent VMEM round trip: ~90 of them per
// ecc_add_unsafe in the value chain (PMC: 47 % of that kernel's cycles in s_waitcnt).
__constant__ H2EFieldConsts g_fc[3];

typedef Wd<2> Limb;   // <= 114 bit
typedef Wd<4> Fe;     // canonical bn256-Fr value

struct InstanceDesc {
    u64* base;          // batch array [base_rows][5][2][n_instances][2] + 2 * instance
    u64* range;         // [range_rows][3][2][n_instances][2] + 2 * instance
    u64* select;        // [select_rows][2][2][n_instances][2] + 2 * instance
    const u64* inputs;  // [n_slots][slot_words]
    u32* status;
    // Workspace of the value chain, instance-minor like the advice arrays: value slot v of this instance sits at
    // ptr + v * ws (ws = n_instances * words per slot; ptr already points at this instance's words of slot 0), so the
    // lanes of a wave - the same slot of consecutive instances - read and write one contiguous run.
    u64* hints;         // [n_hint_slots] quotient / value hints (canonical values)
    u64* nd;            // [n_hint_slots][2] numerator / denominator pairs (Montgomery form) of the V kernels
    u64* jac;           // [n_jac_slots][3] Jacobian scratch of the V kernels
    u64* sel;           // [n_sel_slots][2] points picked by the select pre-kernel (x, y canonical)
    u32 ws;             // words between consecutive value slots
    u32 hs;             // words between the two halves of a cell = 2 x the instances of the array the cell is in: the run's instances, or -
                        // a run made of several caller batches (h2e.h h2e_submit_batches) - those of one batch; the same for every instance
};
// Address spaces.  Pointers that come out of InstanceDesc / H2ELaunch are generic to the compiler, and an access through a
// generic pointer is a FLAT instruction: it counts in the vector-memory AND the LDS counter and completes out of order, so
// every use of an LDS value slot (or of a loaded cell) waited for *all* outstanding global stores of the wave - the
// level-parallel replays paid a store round trip to HBM (~3 k cycles) per round for that.  Every access to cells /
// workspace / inputs therefore goes through g_* (global_load / global_store) and every access to LDS value slots through
// l_* (ds_read / ds_write).
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define H2E_AS_GLOBAL __attribute__((address_space(1)))
#define H2E_AS_LDS __attribute__((address_space(3)))
WI_INLINE u64x2 g_ld16(const u64* p) { return *(const H2E_AS_GLOBAL u64x2*)p; }
WI_INLINE u64 g_ld8(const u64* p) { return *(const H2E_AS_GLOBAL u64*)p; }
WI_INLINE void g_st16(u64* p, u64 x, u64 y) {
    u64x2 v = {x, y};
    *(H2E_AS_GLOBAL u64x2*)p = v;
}
WI_INLINE u64 l_ld8(const u64* p) { return *(const H2E_AS_LDS u64*)p; }
WI_INLINE u64x2 l_ld16(const u64* p) { return *(const H2E_AS_LDS u64x2*)p; }
WI_INLINE void l_st8(u64* p, u64 v) { *(H2E_AS_LDS u64*)p = v; }
WI_INLINE void l_st16(u64* p, u64 x, u64 y) {
    u64x2 v = {x, y};
    *(H2E_AS_LDS u64x2*)p = v;
}
// a W value in a 16-byte aligned workspace slot
template <int N>
WI_INLINE Wd<N> ws_load(const u64* p) {
    Wd<N> r;
#pragma unroll
    for (int i = 0; i < N / 2; i++) {
        u64x2 t = g_ld16(p + 2 * i);
        r.v[2 * i] = t.x;
        r.v[2 * i + 1] = t.y;
    }
    return r;
}
template <int N>
WI_INLINE void ws_store(u64* p, const Wd<N>& v) {
#pragma unroll
    for (int i = 0; i < N / 2; i++) g_st16(p + 2 * i, v.v[2 * i], v.v[2 * i + 1]);
}
// words of the instance inputs / the constant pool (global memory)
template <int N>
WI_INLINE Wd<N> g_load(const u64* p) {
    Wd<N> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = g_ld8(p + i);
    return r;
}

struct LC {  // lane context
    u64* base;
    u64* range;
    u64* select;
    const u64* inputs;
    u32* status;
    u32 ob, orr, os;       // strand offsets
    u32 hs;                // words between the two halves of a cell = 2 * n_instances (2: one instance, row-major)
    const u32* params;     // this strand's parameter refs
    const u32* aux;
    const u64* pool;
    const H2EFieldConsts* fc;
    u32 strand, input_stride;
    u32 sw;                // words per input slot
    const u64* hints;
    u32 hint_stride;
    u32 ws;             // words between consecutive workspace value slots (InstanceDesc)
    const u64* sel;     // selection buffer (H2E_FLAG_PRESELECTED)
    u32 sel_stride;
    bool active;        // false for the padding lanes of the last wave: compute, but store nothing
    // expansion only: three integer results of this lane's sub-range kept in LDS ([3][W][64] words).  An op's
    // operands are mostly the results of the one or two ops before it; re-reading them from their cells misses the L2
    // (19.5 KB written per lane and sub-range against 512 B of L2 per lane) - 22 GB of the window launch's traffic.
    u64* xc = nullptr;
    // wave-mode replay: the find_w_modulus_of_ceil_times tables in LDS ([64][H2E_MAX_L][2] limb words, then [64][4] native
    // words).  Their index is per lane, so from constant memory they are vector loads - and a vector load waits for every
    // older store of the wave (one counter, in order).
    const u64* ceil_lds = nullptr;
    // Stream digest (include/h2e.h h2e_run_digest): the expansion adds every cell it stores to 3 x 4 per-lane sums kept in
    // LDS ([region][word][lane], this lane's words at dg + (4 region + word) * 64), flushed to the run's digest array when
    // the lane is done.  nullptr = off.
    u64* dg = nullptr;
#if H2E_COLS_ON
    // Column emission.  The wave's 64 lanes are 64 consecutive instances (inst0 ..) at the same rows, and a column's cells reach HBM as
    // 128-byte runs of ONE instance - four consecutive rows - because a write request costs the memory system the same whatever it
    // carries up to a line (exp/ubench/colrun.hip, colpolicy.hip: ~43 G requests/s; per-lane 32-byte stores 1.2 TB/s).  So the rows are
    // staged in LDS by blocks of four: base [5 cols][4 rows][64 lanes][32 B] = 40 KB, range compact ([4][64] x 16 B for the
    // accumulator column, x 4 B for the two chunk columns) = 6 KB; a block is flushed - eight lanes per instance and store instruction
    // - when the rows move on.  Unassigned cells inside the rows the sub-range has passed are written as zeros (whole lines; nobody
    // else ever writes an unassigned cell), rows outside are left alone.  52 KB of LDS = three waves per compute unit (+22 % on the
    // window launch; four would be +2 %, two +50 %: exp/r6_occupancy.sh), which leaves each wave a SIMD's registers: the result cache
    // lives in VGPRs here.
    u64* colB = nullptr;   // instance 0's column arrays [col][rows][4 words]
    u64* colR = nullptr;
    u64* colS = nullptr;
    u64 csB = 0, csR = 0, csS = 0;     // words to the next instance's arrays
    u32 crB = 0, crR = 0, crS = 0;     // rows per column
    u32 inst0 = 0;                      // instance of lane 0
    u64* stgB = nullptr;                // LDS
    u64* stgR0 = nullptr;
    u32* stgR12 = nullptr;
    u32 loB = 0, loR = 0;               // first row (absolute) of the sub-range in the base / range array
    mutable u32 blkB = ~0u, validB = 0, hiB = 0;   // the open block (its first row), its staged cells (bit 4 col + row), rows passed
    mutable u32 blkR = ~0u, validR = 0, hiR = 0;
    mutable u64 c3[4][4];               // base column 3 (the rarest: a3 of the four-term rows) is staged in registers - 32 KB + 6 KB + 2 KB of
                                        // LDS = 40 KB let a compute unit take FOUR waves (exp/r6_occupancy.sh: +2 % against +22 % at three)
    mutable u32 dualmask = 0x1fu;       // of the rows an op may read back, the base columns whose cells are handles (limb-wise results: column 4)
    mutable bool nodual = false;        // rows no op ever reads back (the mul equation's): not written to the working copy
    u32 fj = 0, fhalf = 0;              // this lane's piece of a flushed run: row of the block, half of the cell
    u64* fB[8];                         // ... of instance inst0 + 8 s + lane / 8: its base / range column array at that piece (col 0, row 0)
    u64* fR[8];
    u64* lB = nullptr;                  // ... and where the piece sits in the staging (column 0, s = 0)
    u64* lR0 = nullptr;
    u32* lR12 = nullptr;
    mutable u64 xr[3][2 * H2E_MAX_L + 4];
#endif
};
// key of a cell position and the digest contribution of one 64-bit word of its value (h2e.h: stream digest)
WI_INLINE void dg_keys(u32 pos, u32& k0, u32& k1) {
    u32 h = pos * 0x9E3779B1u;
    k0 = (h ^ (h >> 15)) | 1u;
    k1 = (h * 0x85EBCA77u + 0xC2B2AE3Du) | 1u;
}
WI_INLINE void dg_word(u64* slot, u64 w, u32 k0, u32 k1) {
    u64 t = (u64)(u32)w * k0 + (u64)(u32)(w >> 32) * k1;
    __hip_atomic_fetch_add((H2E_AS_LDS u64*)slot, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
// NW = words of the value that can be non-zero (4: any cell, 2: a limb, 1: an 18-bit chunk / small value)
template <int NW>
WI_INLINE void dg_cell(const LC& c, u32 region, u32 abs_row, u32 col, u32 cols, const u64* w) {
    if (c.dg == nullptr) return;
    u32 k0, k1;
    dg_keys(abs_row * cols + col, k0, k1);
    u64* base = c.dg + (size_t)region * 4 * 64;
#pragma unroll
    for (int j = 0; j < NW; j++) dg_word(base + j * 64, w[j], k0, k1);
}
WI_INLINE Limb ceil_limb(const LC& c, u32 t, int i) {
    if (c.ceil_lds) {
        u64x2 q = l_ld16(c.ceil_lds + ((size_t)t * H2E_MAX_L + i) * 2);
        Limb r;
        r.v[0] = q.x;
        r.v[1] = q.y;
        return r;
    }
    return wd_load<2>(c.fc->ceil_limbs[t][i]);
}
WI_INLINE Wd<4> ceil_native(const LC& c, u32 t) {
    if (c.ceil_lds) {
        const u64* p = c.ceil_lds + 64 * H2E_MAX_L * 2 + (size_t)t * 4;
        u64x2 a = l_ld16(p), b = l_ld16(p + 2);
        Wd<4> r;
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b.x; r.v[3] = b.y;
        return r;
    }
    return wd_load<4>(c.fc->ceil_native[t]);
}

// ------------------------------------------------------------------------------------------------
// cell I/O
WI_INLINE u64* cell_ptr(const LC& c, u32 ref) {
    if (H2E_REF_REGION(ref) == H2E_REGION_PARAM) ref = c.params[H2E_REF_ROW(ref)];
    u32 region = H2E_REF_REGION(ref), col = H2E_REF_COL(ref), row = H2E_REF_ROW(ref);
    bool rel = H2E_REF_REL(ref);
    if (region == 0) return c.base + ((size_t)(row + (rel ? c.ob : 0)) * 5 + col) * 2 * c.hs;
    if (region == 1) return c.range + ((size_t)(row + (rel ? c.orr : 0)) * 3 + col) * 2 * c.hs;
    return c.select + ((size_t)(row + (rel ? c.os : 0)) * 2 + col) * 2 * c.hs;
}
WI_INLINE Fe ld_cell(const u64* p, u32 hs) {
    u64x2 a = g_ld16(p), b = g_ld16(p + hs);
    Fe r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b.x; r.v[3] = b.y;
    return r;
}
WI_INLINE Fe ld_fe(const LC& c, u32 ref) { return ld_cell(cell_ptr(c, ref), c.hs); }
WI_INLINE Limb ld_limb(const LC& c, u32 ref) {  // values known to be < 2^128: the low half only
    u64x2 a = g_ld16(cell_ptr(c, ref));
    Limb r;
    r.v[0] = a.x; r.v[1] = a.y;
    return r;
}
// one 16-byte half of a cell.  (Plain stores: non-temporal ones made the window expansion 9 % slower - its operand
// re-reads then miss the L2.)
WI_INLINE void st16(u64* p, u64 x, u64 y) { g_st16(p, x, y); }
WI_INLINE void st_cell(u64* p, u32 hs, const Fe& v) {
    st16(p, v.v[0], v.v[1]);
    st16(p + hs, v.v[2], v.v[3]);
}
WI_INLINE u64* rowB_ptr(const LC& c, u32 row) { return c.base + (size_t)(row + c.ob) * 10 * c.hs; }
WI_INLINE u64* rowR_ptr(const LC& c, u32 row) { return c.range + (size_t)(row + c.orr) * 6 * c.hs; }
WI_INLINE u64* rowS_ptr(const LC& c, u32 row) { return c.select + (size_t)(row + c.os) * 4 * c.hs; }
WI_INLINE void stB(const LC& c, u32 row, int col, const Fe& v) { st_cell(rowB_ptr(c, row) + (size_t)col * 2 * c.hs, c.hs, v); }
WI_INLINE void stR(const LC& c, u32 row, int col, const Fe& v) { st_cell(rowR_ptr(c, row) + (size_t)col * 2 * c.hs, c.hs, v); }
WI_INLINE void stS(const LC& c, u32 row, int col, const Fe& v) { st_cell(rowS_ptr(c, row) + (size_t)col * 2 * c.hs, c.hs, v); }
WI_INLINE Fe fe_of(const Limb& l) { return wd_resize<4>(l); }
WI_INLINE Fe fe_u64(u64 x) { return wd_from_u64<4>(x); }
WI_INLINE void flag(const LC& c, u32 bits) { atomicOr(c.status, bits); }

// ------------------------------------------------------------------------------------------------
// arithmetic mod n (bn256 Fr)
WI_INLINE Fe n_of(const LC& c) { return wd_load<4>(c.fc->n); }
template <int XW>
WI_INLINE Fe mod_n(const LC& c, const Wd<XW>& x) {
    static_assert(XW <= 8, "mod_n input too wide");
    Wd<5> q;
    Fe r;
    wd_barrett_divrem<512, NK, 8, 4, 5>(wd_resize<8>(x), n_of(c), wd_load<5>(c.fc->n_mu), q, r);
    return r;
}
// native (value mod n) of a canonical W element
template <class FP>
WI_INLINE Fe native_of_w(const LC& c, const Wd<FP::WW>& x) {
    if constexpr (FP::NSUB > 0) {
        Fe r = wd_resize<4>(x), n = n_of(c);
#pragma unroll
        for (int i = 0; i < FP::NSUB; i++) r = wd_geq<4>(r, n) ? wd_sub<4>(r, n) : r;
        return r;
    } else {
        return mod_n<FP::WW>(c, x);
    }
}
WI_INLINE Fe addmod_n(const LC& c, const Fe& a, const Fe& b) {
    Fe s = wd_add<4>(a, b);  // < 2n < 2^255
    Fe n = n_of(c);
    return wd_geq<4>(s, n) ? wd_sub<4>(s, n) : s;
}
WI_INLINE Fe submod_n(const LC& c, const Fe& a, const Fe& b) {
    return wd_geq<4>(a, b) ? wd_sub<4>(a, b) : wd_sub<4>(wd_add<4>(a, n_of(c)), b);
}
WI_INLINE Fe mulmod_n(const LC& c, const Fe& a, const Fe& b) { return mod_n<8>(c, wd_mul<4, 4>(a, b)); }
// signed 256-bit (two's complement, |x| < n) -> field element
WI_INLINE Fe fe_of_signed(const LC& c, const Wd<4>& x) { return wd_is_neg<4>(x) ? wd_add<4>(x, n_of(c)) : x; }
WI_INLINE Fe inv_n(const LC& c, const Fe& a) { return wd_inv_mod<4>(a, n_of(c)); }

// ------------------------------------------------------------------------------------------------
// Row emission.  A lane owns whole rows (every row is written by exactly one op) and stores the assigned cells of a
// row straight from registers: the lanes of a wave are consecutive instances at the same (row, col, half), so each
// 16-byte store instruction of the wave is one contiguous 1 KB run (full 128-byte lines, no read-modify-write, no LDS
// staging), and cells the shape leaves unassigned cost nothing.  `mask` = assigned columns (compile-time at nearly
// every call site).  Round 1 kept the reference's per-instance row-major layout: 160 / 96 / 64-byte segments per lane,
// LDS-staged whole-row flushes, 1.48 x the algorithmic bytes written (zero cells) at 0.35 of the HBM roof.
#if H2E_COLS_ON
WI_INLINE u32 uni(u32 x) { return (u32)__builtin_amdgcn_readfirstlane((int)x); }
// flush the open base block: per column with staged cells, eight store instructions of 16 bytes per lane - lane = (instance 8 s +
// lane / 8, piece lane % 8 = row 2 bits, half 1 bit): 128 contiguous bytes per instance.  Every per-lane part of the addresses (the
// instance's array, the piece) is made once per wave (fB / fR / lB ...): per store a uniform offset is added - with the whole
// expression per store the 64-bit multiplications made the flush 400 instructions per row and the launch ALU-bound (17 + 18 ms).
// (called, not inlined: the emission layer has 60 call sites of the row writers, and with the 40 + 24 stores of a flush behind each
// of them the kernel was 137 k instructions - 0.8 MB against a 64 KB instruction cache)
struct ColFlushArgs {
    u32 vb, blk, lo, hi, fj, fhalf, rows;
};
// only3: column 3 alone, bounced through the staging slot of column 0 (free once that column's stores have been issued)
__device__ __attribute__((noinline)) void colB_flush_fn(ColFlushArgs a, u32 only3, const u64* lB, u64* f0, u64* f1, u64* f2, u64* f3, u64* f4, u64* f5, u64* f6,
                                                        u64* f7) {
    u64* const f[8] = {f0, f1, f2, f3, f4, f5, f6, f7};
    const u32 row = a.blk + a.fj;
    const bool own = row >= a.lo && row < a.hi;
    const bool st_on = true;
#pragma unroll
    for (int col = 0; col < 5; col++) {
        if ((col == 3) != (only3 != 0)) continue;
        const u32 vm = (a.vb >> (4 * col)) & 15u;
        if (vm == 0) continue;
        const int slot = col == 3 ? 0 : col == 4 ? 3 : col;
        const bool has = (vm >> a.fj) & 1u;
        const size_t off = ((size_t)col * a.rows + a.blk) * 4;   // wave-uniform
        u64x2 v[8];
#pragma unroll
        for (int s8 = 0; s8 < 8; s8++) {   // (the eight LDS reads as one batch, then the eight stores)
            v[s8].x = 0;
            v[s8].y = 0;
            if (has) v[s8] = l_ld16(lB + (size_t)(slot * 4 * 64 + 8 * s8) * 4);
        }
#pragma unroll
        for (int s8 = 0; s8 < 8; s8++)
            if ((has || own) && st_on) g_st16(f[s8] + off, v[s8].x, v[s8].y);
    }
}
__device__ __attribute__((noinline)) void colR_flush_fn(ColFlushArgs a, const u64* lR0, const u32* lR12, u64* f0, u64* f1, u64* f2, u64* f3, u64* f4, u64* f5, u64* f6,
                                                        u64* f7) {
    u64* const f[8] = {f0, f1, f2, f3, f4, f5, f6, f7};
    const u32 row = a.blk + a.fj;
    const bool own = row >= a.lo && row < a.hi;
    const bool st_on = true;
#pragma unroll
    for (int col = 0; col < 3; col++) {
        const u32 vm = (a.vb >> (4 * col)) & 15u;
        if (vm == 0) continue;
        const bool has = ((vm >> a.fj) & 1u) && a.fhalf == 0;   // (range cells are below 2^128: the high half is zero)
        const bool wr = ((vm >> a.fj) & 1u) || own;
        const size_t off = ((size_t)col * a.rows + a.blk) * 4;
#pragma unroll
        for (int s8 = 0; s8 < 8; s8++) {
            u64x2 v = {0, 0};
            if (has) {
                if (col == 0) v = l_ld16(lR0 + (size_t)(8 * s8) * 2);
                else v.x = (u64) * (const H2E_AS_LDS u32*)(lR12 + (size_t)(col - 1) * 4 * 64 + 8 * s8);
            }
            if (wr && st_on) g_st16(f[s8] + off, v.x, v.y);
        }
    }
}
WI_INLINE void colB_flush(const LC& c, u32 hi) {
    if (c.validB == 0) return;
    ColFlushArgs a = {c.validB, c.blkB, c.loB, hi, c.fj, c.fhalf, c.crB};
    if (c.validB & ~0xf000u) colB_flush_fn(a, 0, c.lB, c.fB[0], c.fB[1], c.fB[2], c.fB[3], c.fB[4], c.fB[5], c.fB[6], c.fB[7]);
    if (c.validB & 0xf000u) {   // column 3: this lane's four rows out of its registers into column 0's slot, then the same flush
        u64* p = c.stgB + (size_t)threadIdx.x * 4;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            l_st16(p + (size_t)j * 64 * 4, c.c3[j][0], c.c3[j][1]);
            l_st16(p + (size_t)j * 64 * 4 + 2, c.c3[j][2], c.c3[j][3]);
        }
        colB_flush_fn(a, 1, c.lB, c.fB[0], c.fB[1], c.fB[2], c.fB[3], c.fB[4], c.fB[5], c.fB[6], c.fB[7]);
    }
    c.validB = 0;
}
WI_INLINE void colR_flush(const LC& c, u32 hi) {
    if (c.validR == 0) return;
    ColFlushArgs a = {c.validR, c.blkR, c.loR, hi, c.fj, c.fhalf, c.crR};
    colR_flush_fn(a, c.lR0, c.lR12, c.fR[0], c.fR[1], c.fR[2], c.fR[3], c.fR[4], c.fR[5], c.fR[6], c.fR[7]);
    c.validR = 0;
}
WI_INLINE void colB_open(const LC& c, u32 arow) {
    if ((arow & ~3u) != c.blkB) {
        if (arow < c.blkB && c.blkB != ~0u) atomicOr(c.status, H2E_STATUS_ARITH);   // rows of a sub-range only move forward
        colB_flush(c, arow);       // (every row below the new one has been passed)
        c.blkB = arow & ~3u;
    }
    c.hiB = arow + 1;
}
WI_INLINE void colR_open(const LC& c, u32 arow) {
    if ((arow & ~3u) != c.blkR) {
        if (arow < c.blkR && c.blkR != ~0u) atomicOr(c.status, H2E_STATUS_ARITH);
        colR_flush(c, arow);
        c.blkR = arow & ~3u;
    }
    c.hiR = arow + 1;
}
WI_INLINE void colB_stage(const LC& c, int col, u32 j, const Fe& v) {
    if (col == 3) {   // j is wave-uniform: a scalar branch per row of the block
        if (j == 0) { _Pragma("unroll") for (int i = 0; i < 4; i++) c.c3[0][i] = v.v[i]; }
        else if (j == 1) { _Pragma("unroll") for (int i = 0; i < 4; i++) c.c3[1][i] = v.v[i]; }
        else if (j == 2) { _Pragma("unroll") for (int i = 0; i < 4; i++) c.c3[2][i] = v.v[i]; }
        else { _Pragma("unroll") for (int i = 0; i < 4; i++) c.c3[3][i] = v.v[i]; }
        return;
    }
    const int slot = col == 4 ? 3 : col;   // LDS staging: columns 0, 1, 2, 4
    u64* p = c.stgB + (((size_t)(slot * 4) + j) * 64 + threadIdx.x) * 4;
    l_st16(p, v.v[0], v.v[1]);
    l_st16(p + 2, v.v[2], v.v[3]);
}
// one range row: which of the three cells are assigned, the accumulator (< 2^128) and the two small values
WI_INLINE void colR_row(const LC& c, u32 arow, u32 mask, u64 a0, u64 a1, u32 tagged, u32 common) {
    colR_open(c, arow);
    const u32 j = arow & 3u;
    if (mask & 1) l_st16(c.stgR0 + ((size_t)j * 64 + threadIdx.x) * 2, a0, a1);
    if (mask & 2) *(H2E_AS_LDS u32*)(c.stgR12 + ((size_t)j) * 64 + threadIdx.x) = tagged;
    if (mask & 4) *(H2E_AS_LDS u32*)(c.stgR12 + ((size_t)4 + j) * 64 + threadIdx.x) = common;
    c.validR |= ((mask & 1) ? 1u << j : 0u) | ((mask & 2) ? 16u << j : 0u) | ((mask & 4) ? 256u << j : 0u);
}
#endif
WI_INLINE void rowB(const LC& c, u32 row, u32 mask, const Fe& v0, const Fe& v1, const Fe& v2, const Fe& v3, const Fe& v4) {
    if (!c.active) return;
#if H2E_COLS_ON
    {
        const u32 arow = uni(row + c.ob);
        colB_open(c, arow);
        const u32 j = arow & 3u;
        if (mask & 1) colB_stage(c, 0, j, v0);
        if (mask & 2) colB_stage(c, 1, j, v1);
        if (mask & 4) colB_stage(c, 2, j, v2);
        if (mask & 8) colB_stage(c, 3, j, v3);
        if (mask & 16) colB_stage(c, 4, j, v4);
        c.validB |= ((mask & 1) ? 1u << j : 0u) | ((mask & 2) ? 16u << j : 0u) | ((mask & 4) ? 256u << j : 0u) | ((mask & 8) ? 4096u << j : 0u) |
                    ((mask & 16) ? 65536u << j : 0u);
        if (c.nodual) return;
        mask &= c.dualmask;
    }
#endif
    u64* p = rowB_ptr(c, row);
    u32 hs = c.hs;
    if (mask & 1) st_cell(p, hs, v0);
    if (mask & 2) st_cell(p + (size_t)2 * hs, hs, v1);
    if (mask & 4) st_cell(p + (size_t)4 * hs, hs, v2);
    if (mask & 8) st_cell(p + (size_t)6 * hs, hs, v3);
    if (mask & 16) st_cell(p + (size_t)8 * hs, hs, v4);
    if (c.dg != nullptr) {
        u32 ar = row + c.ob;
        if (mask & 1) dg_cell<4>(c, 0, ar, 0, 5, v0.v);
        if (mask & 2) dg_cell<4>(c, 0, ar, 1, 5, v1.v);
        if (mask & 4) dg_cell<4>(c, 0, ar, 2, 5, v2.v);
        if (mask & 8) dg_cell<4>(c, 0, ar, 3, 5, v3.v);
        if (mask & 16) dg_cell<4>(c, 0, ar, 4, 5, v4.v);
    }
}
WI_INLINE void rowR(const LC& c, u32 row, u32 mask, const Fe& acc, const Fe& tagged, const Fe& common) {
    if (!c.active) return;
    u64* p = rowR_ptr(c, row);
    u32 hs = c.hs;
#if H2E_COLS_ON
    colR_row(c, uni(row + c.orr), mask, acc.v[0], acc.v[1], (u32)tagged.v[0], (u32)common.v[0]);
    if (!c.nodual && (mask & 1)) st_cell(p, hs, acc);   // (the working copy keeps what an op may read back: accumulator cells - limbs of integers)
    return;
#endif
    if (mask & 1) st_cell(p, hs, acc);
    if (mask & 2) st_cell(p + (size_t)2 * hs, hs, tagged);
    if (mask & 4) st_cell(p + (size_t)4 * hs, hs, common);
    if (c.dg != nullptr) {
        u32 ar = row + c.orr;
        if (mask & 1) dg_cell<4>(c, 1, ar, 0, 3, acc.v);
        if (mask & 2) dg_cell<4>(c, 1, ar, 1, 3, tagged.v);
        if (mask & 4) dg_cell<4>(c, 1, ar, 2, 3, common.v);
    }
}
WI_INLINE void rowS(const LC& c, u32 row, u32 mask, const Fe& value, const Fe& selector) {
    if (!c.active) return;
    u64* p = rowS_ptr(c, row);
    u32 hs = c.hs;
#if H2E_COLS_ON
    {   // the select array (3 % of a tile's cells): straight from the lane, 32 bytes at a time
        const u32 arow = row + c.os;
        u64* q = c.colS + (size_t)(c.inst0 + threadIdx.x) * c.csS + (size_t)arow * 4;
        if ((mask & 1) && arow < c.crS) { g_st16(q, value.v[0], value.v[1]); g_st16(q + 2, value.v[2], value.v[3]); }
        if ((mask & 2) && arow < c.crS) { g_st16(q + (size_t)c.crS * 4, selector.v[0], selector.v[1]); g_st16(q + (size_t)c.crS * 4 + 2, selector.v[2], selector.v[3]); }
    }
#endif
    if (mask & 1) st_cell(p, hs, value);
    if (mask & 2) st_cell(p + (size_t)2 * hs, hs, selector);
    if (c.dg != nullptr) {
        u32 ar = row + c.os;
        if (mask & 1) dg_cell<4>(c, 2, ar, 0, 2, value.v);
        if (mask & 2) dg_cell<4>(c, 2, ar, 1, 2, selector.v);
    }
}
WI_INLINE void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Round barriers of the level / wave / field-chain kernels.  `__syncthreads()` is a workgroup-scope release fence + barrier:
// the fence waits for every outstanding *global* store of the wave (s_waitcnt vmcnt(0)) - a round trip to HBM, ~3 k cycles,
// per round, although the rounds only exchange values through LDS.  These wait for the LDS accesses alone.
WI_INLINE void lds_round_barrier_wave() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }                  // one-wave workgroups
WI_INLINE void lds_round_barrier_workgroup() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
static __device__ const Fe FE0 = {{0, 0, 0, 0}};
// base row with cols 0..k-1 and/or the last column
#define ROW_B1(c, row, a, last) rowB(c, row, 0x11, a, FE0, FE0, FE0, last)
#define ROW_B2(c, row, a, b, last) rowB(c, row, 0x13, a, b, FE0, FE0, last)
#define ROW_B3(c, row, a, b, d, last) rowB(c, row, 0x17, a, b, d, FE0, last)
#define ROW_B4(c, row, a, b, d, e, last) rowB(c, row, 0x1f, a, b, d, e, last)

// ------------------------------------------------------------------------------------------------
// range-chip row groups (src/context.rs:835-972, src/circuit/range_chip.rs:287-347)
WI_INLINE u64 chunk18(const Limb& x, int i) {  // i-th 18-bit chunk of a <=128-bit value
    int sh = 18 * i;
    u64 lo = sh < 64 ? (x.v[0] >> sh) : 0;
    if (sh < 64 && sh + 18 > 64) lo |= x.v[1] << (64 - sh);
    if (sh >= 64) lo = x.v[1] >> (sh - 64);
    return lo & 0x3ffffu;
}
// small values (18-bit chunks, the <= 18-bit common value): the high half of the cell is zero
WI_INLINE void st_small(u64* p, u32 hs, u64 x) {
    st16(p, x, 0);
    st16(p + hs, 0, 0);
}
WI_INLINE void st_limb(u64* p, u32 hs, const Limb& x) {
    st16(p, x.v[0], x.v[1]);
    st16(p + hs, 0, 0);
}
// assign_nonleading_limb: 3 rows, 7 cells (row 0: acc, tagged, common; rows 1, 2: tagged, common; context.rs:909-972)
WI_INLINE void emit_limb3(const LC& c, u32 row, const Limb& x) {
    if (!c.active) return;
    u32 hs = c.hs;
    u64* p = rowR_ptr(c, row);
#if H2E_COLS_ON
    {
        const u32 arow = uni(row + c.orr);
        colR_row(c, arow, 7, x.v[0], x.v[1], (u32)chunk18(x, 3), (u32)chunk18(x, 0));
        colR_row(c, arow + 1, 6, 0, 0, (u32)chunk18(x, 4), (u32)chunk18(x, 1));
        colR_row(c, arow + 2, 6, 0, 0, (u32)chunk18(x, 5), (u32)chunk18(x, 2));
        if (!c.nodual) st_limb(p, hs, x);
        return;
    }
#endif
    size_t cs = (size_t)2 * hs, rs = (size_t)6 * hs;
    st_limb(p, hs, x);
    st_small(p + cs, hs, chunk18(x, 3));
    st_small(p + 2 * cs, hs, chunk18(x, 0));
    st_small(p + rs + cs, hs, chunk18(x, 4));
    st_small(p + rs + 2 * cs, hs, chunk18(x, 1));
    st_small(p + 2 * rs + cs, hs, chunk18(x, 5));
    st_small(p + 2 * rs + 2 * cs, hs, chunk18(x, 2));
    if (c.dg != nullptr) {
        u32 ar = row + c.orr;
        dg_cell<2>(c, 1, ar, 0, 3, x.v);
        const int chunk_of[3][2] = {{3, 0}, {4, 1}, {5, 2}};   // (tagged, common) chunk of each of the three rows
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                u64 v = chunk18(x, chunk_of[r][k]);
                dg_cell<1>(c, 1, ar + r, 1 + k, 3, &v);
            }
    }
}
// leading limb in a 2-line range value (36..72 bits): 2 rows, 5 cells
WI_INLINE void emit_lead2(const LC& c, u32 row, const Limb& x) {
    if (!c.active) return;
    u32 hs = c.hs;
    u64* p = rowR_ptr(c, row);
#if H2E_COLS_ON
    {
        const u32 arow = uni(row + c.orr);
        colR_row(c, arow, 7, x.v[0], x.v[1], (u32)chunk18(x, 2), (u32)chunk18(x, 0));
        colR_row(c, arow + 1, 6, 0, 0, (u32)chunk18(x, 3), (u32)chunk18(x, 1));
        if (!c.nodual) st_limb(p, hs, x);
        return;
    }
#endif
    size_t cs = (size_t)2 * hs, rs = (size_t)6 * hs;
    st_limb(p, hs, x);
    st_small(p + cs, hs, chunk18(x, 2));
    st_small(p + 2 * cs, hs, chunk18(x, 0));
    st_small(p + rs + cs, hs, chunk18(x, 3));
    st_small(p + rs + 2 * cs, hs, chunk18(x, 1));
    if (c.dg != nullptr) {
        u32 ar = row + c.orr;
        dg_cell<2>(c, 1, ar, 0, 3, x.v);
        const int chunk_of[2][2] = {{2, 0}, {3, 1}};
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                u64 v = chunk18(x, chunk_of[r][k]);
                dg_cell<1>(c, 1, ar + r, 1 + k, 3, &v);
            }
    }
}
// assign_common: 1 row, 2 cells
WI_INLINE void emit_common(const LC& c, u32 row, u64 x) {
    if (!c.active) return;
    u64* p = rowR_ptr(c, row);
#if H2E_COLS_ON
    colR_row(c, uni(row + c.orr), 3, x, 0, (u32)x, 0);
    if (!c.nodual) st_small(p, c.hs, x);
    return;
#endif
    st_small(p, c.hs, x);
    st_small(p + (size_t)2 * c.hs, c.hs, x);
    if (c.dg != nullptr) {
        dg_cell<1>(c, 1, row + c.orr, 0, 3, &x);
        dg_cell<1>(c, 1, row + c.orr, 1, 3, &x);
    }
}

template <class FP>
struct IntVal {  // value of an AssignedInteger
    Limb l[FP::L];
    Fe native;
};
template <class FP>
WI_INLINE IntVal<FP> ld_int(const LC& c, const u32* refs) {
    IntVal<FP> r;
#pragma unroll
    for (int i = 0; i < FP::L; i++) r.l[i] = ld_limb(c, refs[i]);
    r.native = ld_fe(c, refs[FP::L]);
    return r;
}
// The host assigns the cache entries per sub-range (furthest-next-use replacement over the static op sequence,
// h2e_capi.cpp assign_expansion_slots): op.flags bits 8-9 = entry + 1 for the result, bits 10-15 for the up to three
// integer operands (0 = cells).
template <class FP>
WI_INLINE IntVal<FP> ld_int_x(const LC& c, const H2EOp& op, int refpos, int which) {
    u32 code = (op.flags >> (10 + 2 * which)) & 3u;
#if H2E_COLS_ON
    if (code != 0) {   // the result cache in registers (a wave has a SIMD's register file to itself here); code is wave-uniform
        IntVal<FP> r;
        const u64* e = code == 1 ? c.xr[0] : code == 2 ? c.xr[1] : c.xr[2];
        u64 w[2 * H2E_MAX_L + 4];
        if (code == 1) { _Pragma("unroll") for (int i = 0; i < 2 * FP::L + 4; i++) w[i] = c.xr[0][i]; }
        else if (code == 2) { _Pragma("unroll") for (int i = 0; i < 2 * FP::L + 4; i++) w[i] = c.xr[1][i]; }
        else { _Pragma("unroll") for (int i = 0; i < 2 * FP::L + 4; i++) w[i] = c.xr[2][i]; }
        (void)e;
#pragma unroll
        for (int i = 0; i < FP::L; i++) {
            r.l[i].v[0] = w[2 * i];
            r.l[i].v[1] = w[2 * i + 1];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) r.native.v[i] = w[2 * FP::L + i];
        return r;
    }
#endif
    if (c.xc != nullptr && code != 0) {
        IntVal<FP> r;
        const u64* p = c.xc + (size_t)(code - 1) * (2 * FP::L + 4) * 64 + threadIdx.x;
#pragma unroll
        for (int i = 0; i < FP::L; i++) {
            r.l[i].v[0] = l_ld8(p + (2 * i) * 64);
            r.l[i].v[1] = l_ld8(p + (2 * i + 1) * 64);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) r.native.v[i] = l_ld8(p + (2 * FP::L + i) * 64);
        return r;
    }
    IntVal<FP> r;
#pragma unroll
    for (int i = 0; i < FP::L; i++) r.l[i] = ld_limb(c, op.refs[refpos + i]);
    r.native = ld_fe(c, op.refs[refpos + FP::L]);
    return r;
}
template <class FP>
WI_INLINE void xc_put_x(const LC& c, const H2EOp& op, const Limb* l, const Fe& native) {
    u32 code = (op.flags >> 8) & 3u;
#if H2E_COLS_ON
    if (code != 0) {
        u64 w[2 * H2E_MAX_L + 4];
#pragma unroll
        for (int i = 0; i < FP::L; i++) {
            w[2 * i] = l[i].v[0];
            w[2 * i + 1] = l[i].v[1];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) w[2 * FP::L + i] = native.v[i];
        if (code == 1) { _Pragma("unroll") for (int i = 0; i < 2 * FP::L + 4; i++) c.xr[0][i] = w[i]; }
        else if (code == 2) { _Pragma("unroll") for (int i = 0; i < 2 * FP::L + 4; i++) c.xr[1][i] = w[i]; }
        else { _Pragma("unroll") for (int i = 0; i < 2 * FP::L + 4; i++) c.xr[2][i] = w[i]; }
    }
    return;
#endif
    if (c.xc == nullptr || code == 0) return;
    u64* p = c.xc + (size_t)(code - 1) * (2 * FP::L + 4) * 64 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < FP::L; i++) {
        l_st8(p + (2 * i) * 64, l[i].v[0]);
        l_st8(p + (2 * i + 1) * 64, l[i].v[1]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) l_st8(p + (2 * FP::L + i) * 64, native.v[i]);
}
// Horner composition of limbs (integer_chip.rs:217-224): sum l_i * 2^(108 i)
template <class FP, int OW>
WI_INLINE Wd<OW> compose(const Limb* l) {
    Wd<OW> r = wd_zero<OW>();
    r = wd_add<OW>(r, wd_resize<OW>(l[0]));
    r = wd_add<OW>(r, wd_shl<OW, 108>(l[1]));
    r = wd_add<OW>(r, wd_shl<OW, 216>(l[2]));
    if (FP::L > 3) r = wd_add<OW>(r, wd_shl<OW, 324>(l[FP::L - 1]));
    return r;
}
// limb i of a canonical value: (x >> 108 i) & (2^108 - 1)
template <int I, int N>
WI_INLINE Limb limb_of(const Wd<N>& x) {
    return wd_mask<108, 2>(wd_shr<2, 108 * I>(x));
}
template <class FP, int N>
WI_INLINE void split_limbs(const Wd<N>& x, Limb* out) {
    out[0] = limb_of<0>(x);
    out[1] = limb_of<1>(x);
    out[2] = limb_of<2>(x);
    if (FP::L > 3) out[FP::L - 1] = limb_of<3>(x);
}
// base row [limb_0 .. limb_{L-1} | last]
template <class FP>
WI_INLINE void row_limbs(const LC& c, u32 row, const Limb* l, const Fe& last) {
    if (FP::L == 3)
        ROW_B3(c, row, fe_of(l[0]), fe_of(l[1]), fe_of(l[2]), last);
    else
        ROW_B4(c, row, fe_of(l[0]), fe_of(l[1]), fe_of(l[2]), fe_of(l[FP::L - 1]), last);
}

// assign_w / assign_d body: range rows for the limbs + the base row [limbs .. | native]
// (integer_chip.rs:236-281).  Returns rows consumed in range.
template <class FP>
WI_INLINE u32 emit_assigned(const LC& c, u32 brow, u32 rrow, const Limb* l, const Fe& native) {
    u32 r = rrow;
#pragma unroll
    for (int i = 0; i < FP::L - 1; i++) {
        emit_limb3(c, r, l[i]);
        r += 3;
    }
    emit_lead2(c, r, l[FP::L - 1]);
    r += 2;
#if H2E_COLS_ON
    const u32 dm_was = c.dualmask;
    c.dualmask = 0x10u;   // (the limbs' handles are the range cells; of this row only the native is read back)
#endif
    row_limbs<FP>(c, brow, l, native);
#if H2E_COLS_ON
    c.dualmask = dm_was;
#endif
    return r - rrow;
}

template <class FP>
WI_INLINE void divrem_w(const LC& c, const Wd<FPX<FP>::XW>& X, Wd<FPX<FP>::QW>& q, Wd<FP::WW>& r) {
    wd_barrett_divrem<FPX<FP>::S, FP::K, FPX<FP>::XW, FP::WW, FPX<FP>::QW>(X, wd_load<FP::WW>(c.fc->w),
                                                                             wd_load<FPX<FP>::QW>(c.fc->w_mu), q, r);
}

// floor division of a composed operand A < 2^(CEIL+6) by w: the quotient is < 2^7, so estimate it from the top
// 64 bits and correct (reduce: integer_chip.rs:296-297; also the canonical representatives in int_div).
template <class FP>
WI_INLINE void divrem_small(const LC& c, const Wd<FPX<FP>::AW>& A, u64& q, Wd<FP::WW>& r) {
    constexpr int AW = FPX<FP>::AW;
    constexpr int SH = FP::K - 57;                    // keep 57 significant bits of w
    Wd<AW> w = wd_resize<AW>(wd_load<FP::WW>(c.fc->w));
    u64 a_top = wd_shr<1, SH>(A).v[0];               // < 2^(CEIL + 6 - K + 57) <= 2^63
    u64 w_top = wd_shr<1, SH>(w).v[0];               // in [2^56, 2^57)
    u64 qe = a_top / (w_top + 1);                     // never over-estimates; under-estimates by at most 1
    Wd<AW> rr = wd_sub<AW>(A, wd_resize<AW>(wd_mul<AW, 1>(w, wd_from_u64<1>(qe))));
#pragma unroll
    for (int it = 0; it < 2; it++) {
        bool ge = wd_geq<AW>(rr, w);
        rr = wd_select<AW>(ge, wd_sub<AW>(rr, w), rr);
        qe += ge ? 1 : 0;
    }
    q = qe;
    r = wd_resize<FP::WW>(rr);
}

// limb product as a signed 256-bit value
WI_INLINE Wd<4> lmul(const Limb& a, const Limb& b) { return wd_mul<2, 2>(a, b); }

// The mul-equation rows (integer_chip.rs:73-215) for a*b = d*w + rem, with rem/d already assigned.
// Writes base rows from `brow` and range rows from `rrow`.
template <class FP>
WI_INLINE void emit_mul_equation(const LC& c, u32 brow, u32 rrow, const IntVal<FP>& a, const IntVal<FP>& b,
                                 const Limb* d, const Fe& d_native, const Limb* rem, const Fe& rem_native) {
    constexpr int L = FP::L;
#if H2E_COLS_ON
    const bool nodual_was = c.nodual;   // no op reads these rows back
    c.nodual = true;
#endif
    Wd<4> lv[FP::MC];
    Limb wl[L];
#pragma unroll
    for (int i = 0; i < L; i++) wl[i] = wd_load<2>(c.fc->w_limbs[i]);
    // cross-product rows (mul_add_with_next_line, base_chip.rs:245-281)
#pragma unroll
    for (int pos = 0; pos < FP::MC; pos++) {
        const int r_bound = (pos + 1 < L) ? pos + 1 : L;
        const int l_bound = (pos >= L - 1) ? pos - (L - 1) : 0;
        Wd<4> t = wd_zero<4>();
        if (r_bound - l_bound == 1) {
            const int i = l_bound;
            t = wd_sub<4>(lmul(a.l[i], b.l[pos - i]), lmul(d[i], wl[pos - i]));
            ROW_B3(c, brow, fe_of(a.l[i]), fe_of(b.l[pos - i]), fe_of(d[i]), fe_of_signed(c, t));
            brow += 1;
        } else {
#pragma unroll
            for (int i = l_bound; i < r_bound; i++) {
                ROW_B3(c, brow, fe_of(a.l[i]), fe_of(b.l[pos - i]), fe_of(d[i]), fe_of_signed(c, t));
                brow += 1;
                t = wd_add<4>(t, wd_sub<4>(lmul(a.l[i], b.l[pos - i]), lmul(d[i], wl[pos - i])));
            }
            rowB(c, brow, 0x10, FE0, FE0, FE0, FE0, fe_of_signed(c, t));
            brow += 1;
        }
        lv[pos] = t;
    }
    // borrow = L * 2^108 + 2 ; K0 = 2^108 * borrow ; K1 = K0 - borrow     (integer_chip.rs:112-145)
    Wd<4> borrow = wd_add<4>(wd_shl<4, 108>(wd_from_u64<1>((u64)L)), wd_from_u64<4>(2));
    Wd<4> K0 = wd_shl<4, 108>(borrow);
    Wd<4> K1 = wd_sub<4>(K0, borrow);
    Limb v_l = wd_zero<2>();
    u64 v_h = 0;
#pragma unroll
    for (int i = 0; i < FP::MC; i++) {
        Wd<4> u;
        if (i == 0) {
            u = wd_add<4>(wd_sub<4>(lv[0], wd_resize<4>(rem[0])), K0);
            ROW_B2(c, brow, fe_of_signed(c, lv[0]), fe_of(rem[0]), u);
        } else if (i < L) {
            Wd<4> vprev = wd_add<4>(wd_shl<4, 108>(wd_from_u64<1>(v_h)), wd_resize<4>(v_l));
            u = wd_add<4>(wd_add<4>(wd_sub<4>(lv[i], wd_resize<4>(rem[i])), vprev), K1);
            ROW_B4(c, brow, fe_of_signed(c, lv[i]), fe_of(rem[i]), fe_u64(v_h), fe_of(v_l), u);
        } else {
            Wd<4> vprev = wd_add<4>(wd_shl<4, 108>(wd_from_u64<1>(v_h)), wd_resize<4>(v_l));
            u = wd_add<4>(wd_add<4>(lv[i], vprev), K1);
            ROW_B3(c, brow, fe_of_signed(c, lv[i]), fe_u64(v_h), fe_of(v_l), u);
        }
        brow += 1;
        if (wd_is_neg<4>(u) || !wd_is_zero<2>(wd_mask<108, 2>(wd_resize<2>(u)))) flag(c, H2E_STATUS_ARITH);
        Wd<4> v = wd_shr<4, 108>(u);
        v_l = wd_mask<108, 2>(wd_resize<2>(v));
        v_h = wd_shr<1, 108>(v).v[0];
        emit_common(c, rrow, v_h);
        emit_limb3(c, rrow + 1, v_l);
        rrow += 4;
        ROW_B2(c, brow, fe_u64(v_h), fe_of(v_l), u);
        brow += 1;
    }
    // native row (integer_chip.rs:195-215)
    rowB(c, brow, 0x0f, a.native, b.native, d_native, rem_native, FE0);
#if H2E_COLS_ON
    c.nodual = nodual_was;
#endif
}

// ------------------------------------------------------------------------------------------------
// ops
template <class FP>
WI_INLINE void op_assign_w(const LC& c, const H2EOp& op) {
    u32 slot = op.imm + ((op.flags & H2E_FLAG_INPUT_STRIDED) ? c.strand * c.input_stride : 0);
    Wd<FP::WW> x = g_load<FP::WW>(c.inputs + (size_t)slot * c.sw);
    Limb l[FP::L];
    split_limbs<FP>(x, l);
    emit_assigned<FP>(c, op.base_row, op.range_row, l, native_of_w<FP>(c, x));
}

template <class FP>
WI_INLINE void op_const_int(const LC& c, const H2EOp& op, bool from_input) {
    Wd<FP::WW> x;
    if (from_input) {
        u32 slot = op.imm + ((op.flags & H2E_FLAG_INPUT_STRIDED) ? c.strand * c.input_stride : 0);
        x = g_load<FP::WW>(c.inputs + (size_t)slot * c.sw);
    } else {
        x = g_load<FP::WW>(c.pool + op.imm);
    }
    Limb l[FP::L];
    split_limbs<FP>(x, l);
#pragma unroll
    for (int i = 0; i < FP::L; i++) rowB(c, op.base_row + i, 1, fe_of(l[i]), FE0, FE0, FE0, FE0);
    rowB(c, op.base_row + FP::L, 1, mod_n<FP::WW>(c, x), FE0, FE0, FE0, FE0);
}

template <class FP>
WI_INLINE void op_int_add(const LC& c, const H2EOp& op) {
#if H2E_COLS_ON
    c.dualmask = 0x10u;   // (limb-wise op: the result's handles are the column-4 cells of its rows; the other columns hold copies)
#endif
    IntVal<FP> a = ld_int_x<FP>(c, op, 0, 0), b = ld_int_x<FP>(c, op, FP::L + 1, 1);
    u32 r = op.base_row;
    Limb s[FP::L];
#pragma unroll
    for (int i = 0; i < FP::L; i++) {
        s[i] = wd_add<2>(a.l[i], b.l[i]);
        ROW_B2(c, r + i, fe_of(a.l[i]), fe_of(b.l[i]), fe_of(s[i]));
    }
    Fe nat = addmod_n(c, a.native, b.native);
    row_limbs<FP>(c, r + FP::L, s, nat);
    xc_put_x<FP>(c, op, s, nat);
#if H2E_COLS_ON
    c.dualmask = 0x1fu;
#endif
}

template <class FP>
WI_INLINE void op_int_sub(const LC& c, const H2EOp& op) {
#if H2E_COLS_ON
    c.dualmask = 0x10u;   // (limb-wise op: the result's handles are the column-4 cells of its rows; the other columns hold copies)
#endif
    IntVal<FP> a = ld_int_x<FP>(c, op, 0, 0), b = ld_int_x<FP>(c, op, FP::L + 1, 1);
    u32 r = op.base_row, t = op.imm;
    Limb s[FP::L];
#pragma unroll
    for (int i = 0; i < FP::L; i++) {
        Limb U = wd_load<2>(c.fc->ceil_limbs[t][i]);
        s[i] = wd_sub<2>(wd_add<2>(a.l[i], U), b.l[i]);
        ROW_B2(c, r + i, fe_of(a.l[i]), fe_of(b.l[i]), fe_of(s[i]));
    }
    Fe un = wd_load<4>(c.fc->ceil_native[t]);
    Fe nat = addmod_n(c, submod_n(c, a.native, b.native), un);
    row_limbs<FP>(c, r + FP::L, s, nat);
    xc_put_x<FP>(c, op, s, nat);
#if H2E_COLS_ON
    c.dualmask = 0x1fu;
#endif
}

template <class FP>
WI_INLINE void op_int_neg(const LC& c, const H2EOp& op) {
#if H2E_COLS_ON
    c.dualmask = 0x10u;   // (limb-wise op: the result's handles are the column-4 cells of its rows; the other columns hold copies)
#endif
    IntVal<FP> a = ld_int_x<FP>(c, op, 0, 0);
    u32 r = op.base_row, t = op.imm;
    Limb s[FP::L];
#pragma unroll
    for (int i = 0; i < FP::L; i++) {
        Limb U = wd_load<2>(c.fc->ceil_limbs[t][i]);
        s[i] = wd_sub<2>(U, a.l[i]);
        ROW_B1(c, r + i, fe_of(a.l[i]), fe_of(s[i]));
    }
    Fe un = wd_load<4>(c.fc->ceil_native[t]);
    Fe nat = submod_n(c, un, a.native);
    row_limbs<FP>(c, r + FP::L, s, nat);
    xc_put_x<FP>(c, op, s, nat);
#if H2E_COLS_ON
    c.dualmask = 0x1fu;
#endif
}

template <class FP>
WI_INLINE void op_int_mul_small(const LC& c, const H2EOp& op) {
#if H2E_COLS_ON
    c.dualmask = 0x10u;   // (limb-wise op: the result's handles are the column-4 cells of its rows; the other columns hold copies)
#endif
    IntVal<FP> a = ld_int_x<FP>(c, op, 0, 0);
    u32 r = op.base_row;
    Wd<1> k = wd_from_u64<1>(op.imm);
    Limb s[FP::L];
#pragma unroll
    for (int i = 0; i < FP::L; i++) {
        s[i] = wd_resize<2>(wd_mul<2, 1>(a.l[i], k));
        ROW_B1(c, r + i, fe_of(a.l[i]), fe_of(s[i]));
    }
    Fe nat = mod_n<5>(c, wd_mul<4, 1>(a.native, k));
    row_limbs<FP>(c, r + FP::L, s, nat);
    xc_put_x<FP>(c, op, s, nat);
#if H2E_COLS_ON
    c.dualmask = 0x1fu;
#endif
}

// A hinted INT_MUL / REDUCE: the values-only replay took the result from the hint slot, so every later op was fed
// that value; the expansion computes the real one and a difference must not go unnoticed.
template <class FP>
WI_INLINE u32 hint_slot_of(const LC& c, const H2EOp& op) {
    return op.imm + ((op.flags & H2E_FLAG_HINT_STRIDED) ? c.strand * c.hint_stride : 0);
}
template <class FP>
WI_INLINE void check_value_hint(const LC& c, const H2EOp& op, const Wd<FP::WW>& rem) {
    if (op.flags & H2E_FLAG_HINTED) {
        Wd<FP::WW> h = ws_load<FP::WW>(c.hints + (size_t)hint_slot_of<FP>(c, op) * c.ws);
        if (!wd_eq<FP::WW>(h, rem)) flag(c, H2E_STATUS_ARITH);
    }
}
template <class FP>
WI_INLINE void op_int_mul(const LC& c, const H2EOp& op) {
    constexpr int L = FP::L;
    IntVal<FP> a = ld_int_x<FP>(c, op, 0, 0), b = ld_int_x<FP>(c, op, L + 1, 1);
    Wd<FPX<FP>::AW> A = compose<FP, FPX<FP>::AW>(a.l), B = compose<FP, FPX<FP>::AW>(b.l);
    Wd<FPX<FP>::XW> X = wd_resize<FPX<FP>::XW>(wd_mul<FPX<FP>::AW, FPX<FP>::AW, FPX<FP>::AL, FPX<FP>::AL>(A, B));
    Wd<FPX<FP>::QW> dq;
    Wd<FP::WW> rem;
    divrem_w<FP>(c, X, dq, rem);
    check_value_hint<FP>(c, op, rem);
    Limb rl[L], dl[L];
    split_limbs<FP>(rem, rl);
    split_limbs<FP>(dq, dl);
    Fe rem_native = native_of_w<FP>(c, rem), d_native = mod_n<FPX<FP>::QW>(c, dq);
    u32 rr = op.range_row;
    xc_put_x<FP>(c, op, rl, rem_native);
    rr += emit_assigned<FP>(c, op.base_row, rr, rl, rem_native);
#if H2E_COLS_ON
    c.nodual = true;    // (the quotient's cells are nobody's operand)
#endif
    rr += emit_assigned<FP>(c, op.base_row + 1, rr, dl, d_native);
    emit_mul_equation<FP>(c, op.base_row + 2, rr, a, b, dl, d_native, rl, rem_native);
#if H2E_COLS_ON
    c.nodual = false;
#endif
}

template <class FP>
WI_INLINE void op_reduce(const LC& c, const H2EOp& op) {
    constexpr int L = FP::L;
    IntVal<FP> a = ld_int_x<FP>(c, op, 0, 0);
    Wd<FPX<FP>::AW> A = compose<FP, FPX<FP>::AW>(a.l);
    Wd<FP::WW> rem;
    u64 d;
    divrem_small<FP>(c, A, d, rem);
    check_value_hint<FP>(c, op, rem);
    Limb rl[L];
    split_limbs<FP>(rem, rl);
    Fe rem_native = native_of_w<FP>(c, rem);
    u32 rr = op.range_row, br = op.base_row;
    xc_put_x<FP>(c, op, rl, rem_native);
    rr += emit_assigned<FP>(c, br, rr, rl, rem_native);
#if H2E_COLS_ON
    c.nodual = true;   // quotient and carries: nobody's operands
#endif
    emit_common(c, rr, d);
    rr += 1;
    // native row: [d * w_native, rem.native * 1 | a.native * (-1)]   (integer_chip.rs:303-311)
    ROW_B2(c, br + 1, fe_u64(d), rem_native, a.native);
    br += 2;
    Limb last_v = wd_zero<2>();
#pragma unroll
    for (int i = 0; i < FP::RC; i++) {
        // u = d*w_i + rem_i + 64*2^108 - a_i + carry - (i ? 64 : 0)     (integer_chip.rs:332-346)
        Limb wl = wd_load<2>(c.fc->w_limbs[i]);
        Wd<4> u = wd_resize<4>(wd_mul<2, 1>(wl, wd_from_u64<1>(d)));
        u = wd_add<4>(u, wd_resize<4>(rl[i]));
        u = wd_add<4>(u, wd_shl<4, 108>(wd_from_u64<1>(64)));
        u = wd_sub<4>(u, wd_resize<4>(a.l[i]));
        u = wd_add<4>(u, wd_resize<4>(last_v));
        if (i != 0) u = wd_sub<4>(u, wd_from_u64<4>(64));
        if (wd_is_neg<4>(u) || !wd_is_zero<2>(wd_mask<108, 2>(wd_resize<2>(u)))) flag(c, H2E_STATUS_ARITH);
        Limb v = wd_resize<2>(wd_shr<4, 108>(u));
        emit_limb3(c, rr, v);
        rr += 3;
        // 4th operand is pair!(zero, zero) for i == 0 (quirk Q5): the cell is assigned, value 0
        ROW_B4(c, br, fe_u64(d), fe_of(rl[i]), fe_of(a.l[i]), fe_of(last_v), fe_of(v));
        br += 1;
        last_v = v;
    }
#if H2E_COLS_ON
    c.nodual = false;
#endif
}

// invert rows for one value (base_chip.rs:298-321): [a, c] ; [a, b | c].  b = a^-1 (row + 1, col 1) is left to
// the fix-up kernel.
WI_INLINE void emit_invert(const LC& c, u32 row, const Fe& a) {
    Fe cc = fe_u64(wd_is_zero<4>(a) ? 1 : 0);
    rowB(c, row, 0x03, a, cc, FE0, FE0, FE0);
    rowB(c, row + 1, 0x11, a, FE0, FE0, FE0, cc);
}

template <class FP>
WI_INLINE void op_is_int_zero(const LC& c, const H2EOp& op) {
    constexpr int L = FP::L;
    IntVal<FP> a = ld_int_x<FP>(c, op, 0, 0);
    Limb sum = a.l[0];
#pragma unroll
    for (int i = 1; i < L; i++) sum = wd_add<2>(sum, a.l[i]);
    Fe x0 = fe_of(sum);
    Fe x1 = submod_n(c, a.native, wd_load<4>(c.fc->w_native));
    u32 r = op.base_row;
    // is_pure_zero (integer_chip.rs:540-548)
    row_limbs<FP>(c, r, a.l, x0);
    emit_invert(c, r + 1, x0);
    u64 is_zero = wd_is_zero<4>(x0) ? 1 : 0;
    r += 3;
    // is_pure_w_modulus (integer_chip.rs:550-570)
    ROW_B1(c, r, a.native, x1);
    emit_invert(c, r + 1, x1);
    u64 is_eq = wd_is_zero<4>(x1) ? 1 : 0;
    r += 3;
#pragma unroll
    for (int i = 0; i < FP::PW; i++) {
        Fe xi = submod_n(c, fe_of(a.l[i]), fe_of(wd_load<2>(c.fc->w_limbs[i])));
        ROW_B1(c, r, fe_of(a.l[i]), xi);
        emit_invert(c, r + 1, xi);
        u64 is_limb_eq = wd_is_zero<4>(xi) ? 1 : 0;
        ROW_B2(c, r + 3, fe_u64(is_eq), fe_u64(is_limb_eq), fe_u64(is_eq & is_limb_eq));
        is_eq &= is_limb_eq;
        r += 4;
    }
    // or (base_chip.rs:428-439)
    ROW_B2(c, r, fe_u64(is_zero), fe_u64(is_eq), fe_u64(is_zero | is_eq));
}

template <class FP>
WI_INLINE void op_mask_int(const LC& c, const H2EOp& op) {
    constexpr int L = FP::L;
    IntVal<FP> a = ld_int_x<FP>(c, op, 0, 0);
    Fe coeff = ld_fe(c, op.refs[L + 1]);
    bool keep = !wd_is_zero<4>(coeff);
    u32 r = op.base_row;
#pragma unroll
    for (int i = 0; i < L; i++) ROW_B2(c, r + i, fe_of(a.l[i]), coeff, keep ? fe_of(a.l[i]) : wd_zero<4>());
    ROW_B2(c, r + L, a.native, coeff, keep ? a.native : wd_zero<4>());
}

// int_div core (integer_chip.rs:522-535): refs = b (L+1), a' (L+1)
template <class FP>
WI_INLINE void op_div_core(const LC& c, const H2EOp& op) {
    constexpr int L = FP::L;
    IntVal<FP> b = ld_int_x<FP>(c, op, 0, 0), a = ld_int_x<FP>(c, op, L + 1, 1);
    Wd<FPX<FP>::AW> A = compose<FP, FPX<FP>::AW>(a.l), B = compose<FP, FPX<FP>::AW>(b.l);
    Wd<FP::WW> w = wd_load<FP::WW>(c.fc->w);
    // canonical representatives mod w (bn_to_field::<W>)
    Wd<FPX<FP>::QW> q0;
    Wd<FP::WW> a_red, b_red;
    u64 qs;
    divrem_small<FP>(c, A, qs, a_red);
    divrem_small<FP>(c, B, qs, b_red);
    Wd<FP::WW> cv;
    if (op.flags & H2E_FLAG_HINTED) {
        // c = a * b^-1 mod w was predicted by the V kernels (native Montgomery arithmetic + batch inversion);
        // a wrong hint cannot go unnoticed: (b*c - a) must be an exact multiple of w below.
        u32 slot = op.imm + ((op.flags & H2E_FLAG_HINT_STRIDED) ? c.strand * c.hint_stride : 0);
        cv = ws_load<FP::WW>(c.hints + (size_t)slot * c.ws);
        if (wd_is_zero<FP::WW>(b_red)) cv = wd_zero<FP::WW>();
    } else {
        Wd<FP::WW> binv = wd_inv_mod<FP::WW>(b_red, w);
        divrem_w<FP>(c, wd_resize<FPX<FP>::XW>(wd_mul<FP::WW, FP::WW>(a_red, binv)), q0, cv);
    }
    // d = (b*c - a) / w   (exact)
    Wd<FPX<FP>::XW> bc = wd_resize<FPX<FP>::XW>(wd_mul<FPX<FP>::AW, FP::WW>(B, cv));
    Wd<FPX<FP>::XW> num = wd_sub<FPX<FP>::XW>(bc, wd_resize<FPX<FP>::XW>(A));
    Wd<FPX<FP>::QW> dq;
    Wd<FP::WW> rz;
    divrem_w<FP>(c, num, dq, rz);
    if (wd_is_neg<FPX<FP>::XW>(num) || !wd_is_zero<FP::WW>(rz)) flag(c, H2E_STATUS_ARITH);
    Limb cl[L], dl[L];
    split_limbs<FP>(cv, cl);
    split_limbs<FP>(dq, dl);
    Fe c_native = native_of_w<FP>(c, cv), d_native = mod_n<FPX<FP>::QW>(c, dq);
    u32 rr = op.range_row;
    xc_put_x<FP>(c, op, cl, c_native);
    rr += emit_assigned<FP>(c, op.base_row, rr, cl, c_native);
#if H2E_COLS_ON
    c.nodual = true;
#endif
    rr += emit_assigned<FP>(c, op.base_row + 1, rr, dl, d_native);
    IntVal<FP> cvv;
#pragma unroll
    for (int i = 0; i < L; i++) cvv.l[i] = cl[i];
    cvv.native = c_native;
    emit_mul_equation<FP>(c, op.base_row + 2, rr, b, cvv, dl, d_native, a.l, a.native);
#if H2E_COLS_ON
    c.nodual = false;
#endif
}

template <class FP>
WI_INLINE void op_bisec_int(const LC& c, const H2EOp& op) {
    // imm = limbs of the integers (0: this kernel's field).  A GeneralScalarEccContext bisects scalars of its other
    // integer context (3 limbs) inside a fork of the base field (4 limbs): the rows are base-chip rows only.
    const int L = op.imm ? (int)op.imm : FP::L;
    Fe cond = ld_fe(c, op.refs[0]);
    bool take_a = !wd_is_zero<4>(cond);
    u32 r = op.base_row;
    static_assert(H2E_MAX_L == 4 && H2E_OP_MAX_REFS >= 11, "b's references start at 5 (three limbs) or 6 (four)");
#pragma unroll
    for (int i = 0; i <= H2E_MAX_L; i++) {
        if (i <= L) {
            // (constant indices only: where the op is per-lane data - the packed expansion - a runtime index would send the whole
            // reference array to scratch memory)
            const u32 rb = L == 3 ? op.refs[i + 5 <= 10 ? i + 5 : 10] : op.refs[i + 6 <= 10 ? i + 6 : 10];
            Fe av = ld_fe(c, op.refs[1 + i]), bv = ld_fe(c, rb);
            ROW_B4(c, r + i, cond, av, cond, bv, take_a ? av : bv);
        }
    }
}

// one limb of GeneralScalarEccContext::decompose_scalar::<1> (general_scalar_ecc_chip.rs:107-130): imm = limb bits
WI_INLINE void op_decompose_limb(const LC& c, const H2EOp& op) {
    Limb rest = ld_limb(c, op.refs[0]);
    u32 r = op.base_row, nbits = op.imm;
    for (u32 j = 0; j < nbits; j++) {
        u64 b = rest.v[0] & 1;
        Limb v = wd_shr1<2>(rest);
        rowB(c, r, 3, fe_u64(b), fe_u64(b), FE0, FE0, FE0);                 // assign_bit: two copies (quirk Q2)
        ROW_B2(c, r + 1, fe_of(rest), fe_u64(b), fe_of(v));                 // [rest * -1, b * 1 | v * 2]
        r += 2;
        rest = v;
    }
    rowB(c, r, 1, fe_of(rest), FE0, FE0, FE0, FE0);                         // assert_constant(rest, 0)
    if (c.active && !wd_is_zero<2>(rest)) flag(c, H2E_STATUS_ASSERT_FAILED);
}

template <class FP>
WI_INLINE void op_sum_limbs(const LC& c, const H2EOp& op) {
    Limb l[FP::L];
    Limb sum = wd_zero<2>();
#pragma unroll
    for (int i = 0; i < FP::L; i++) {
        l[i] = ld_limb(c, op.refs[i]);
        sum = wd_add<2>(sum, l[i]);
    }
    row_limbs<FP>(c, op.base_row, l, fe_of(sum));
}

WI_INLINE void op_assert_const(const LC& c, const H2EOp& op) {
    Fe x = ld_fe(c, op.refs[0]);
    rowB(c, op.base_row, 1, x, FE0, FE0, FE0, FE0);
    if (c.active && !wd_eq<4>(x, fe_u64(op.imm))) {
        u32 bits = H2E_STATUS_ASSERT_FAILED;
        if (op.flags & H2E_FLAG_UNSAFE_ADD) bits |= H2E_STATUS_RETRY_ADD_SAME_OR_NEG;
        if (op.flags & H2E_FLAG_UNSAFE_DBL) bits |= H2E_STATUS_RETRY_ADD_IDENTITY;
        flag(c, bits);
    }
}

// native decompose_scalar::<1> (native_scalar_ecc_chip.rs:97-171): NUM_BITS/2 x [bit, bit, recombine] + assert
WI_INLINE void op_decompose_native(const LC& c, const H2EOp& op) {
    Fe s = ld_fe(c, op.refs[0]);
    u32 r = op.base_row;
    u32 nbits = op.imm;
    Fe v = s;
    for (u32 i = 0; i < nbits / 2; i++) {
        u64 b0 = v.v[0] & 1, b1 = (v.v[0] >> 1) & 1;
        Fe vn = wd_shr1<4>(wd_shr1<4>(v));
        rowB(c, r, 3, fe_u64(b0), fe_u64(b0), FE0, FE0, FE0);
        rowB(c, r + 1, 3, fe_u64(b1), fe_u64(b1), FE0, FE0, FE0);
        ROW_B3(c, r + 2, vn, fe_u64(b1), fe_u64(b0), v);
        r += 3;
        v = vn;
    }
    // even NUM_BITS: assert_constant(v, 0); odd: assert_bit(v)
    if (nbits & 1) {
        rowB(c, r, 3, v, v, FE0, FE0, FE0);
    } else {
        rowB(c, r, 1, v, FE0, FE0, FE0, FE0);
        if (c.active && !wd_is_zero<4>(v)) flag(c, H2E_STATUS_ASSERT_FAILED);
    }
}

// index = sum bit_i * 2^i over imm (<= 5) bit cells (ecc_chip.rs:941-948; sum_with_constant splits at 5 terms)
WI_INLINE void op_pick_index(const LC& c, const H2EOp& op) {
    u32 k = op.imm, r = op.base_row;
    u64 bits[5];
#pragma unroll
    for (int i = 0; i < 5; i++) bits[i] = (i < (int)k) ? ld_fe(c, op.refs[i]).v[0] : 0;
    u64 acc = bits[0] | (bits[1] << 1) | (bits[2] << 2) | (bits[3] << 3);
    if (k < 5) {
        rowB(c, r, ((1u << k) - 1) | 0x10, fe_u64(bits[0]), fe_u64(bits[1]), fe_u64(bits[2]), fe_u64(bits[3]), fe_u64(acc));
    } else {
        ROW_B4(c, r, fe_u64(bits[0]), fe_u64(bits[1]), fe_u64(bits[2]), fe_u64(bits[3]), fe_u64(acc));
        ROW_B2(c, r + 1, fe_u64(bits[4]), fe_u64(acc), fe_u64(acc | (bits[4] << 4)));
    }
}

template <class FP>
WI_INLINE void op_cache_int(const LC& c, const H2EOp& op) {
    const int n = op.imm ? (int)op.imm : FP::L + 1;   // imm: cells to cache (single cells of assign_cache_point, ecc_chip.rs:779-788)
#pragma unroll
    for (int i = 0; i <= H2E_MAX_L; i++)
        if (i < n) rowS(c, op.select_row + i, 1, ld_fe(c, op.refs[i]), FE0);
}

// sum_with_constant([(a, 1), (b, 2^108)], None) of ecc_encode (ecc_chip.rs:718-731): [a, b | a + b * 2^108]
WI_INLINE void op_shift_add(const LC& c, const H2EOp& op) {
    Fe a = ld_fe(c, op.refs[0]), b = ld_fe(c, op.refs[1]);
    Fe s = mod_n<6>(c, wd_add<6>(wd_resize<6>(a), wd_shl<6, 108>(b)));
    ROW_B2(c, op.base_row, a, b, s);
}

// assign_selected_point_non_zero: refs[0] = index cell, imm = aux offset of the candidate ref table
// [n_candidates][2*(L+1)] (ecc_chip.rs:949-967)
template <class FP>
WI_INLINE void op_select_point(const LC& c, const H2EOp& op) {
    constexpr int NC = 2 * (FP::L + 1);
    Fe index = ld_fe(c, op.refs[0]);
    u32 idx = (u32)(index.v[0] & 0xff);
    u32 nc_dyn = (op.flags >> 8) & 0xffu;
    if (nc_dyn) {   // candidates of another width (points with curvature, ecc_chip.rs:790-812): one cell at a time
        const u32* tab = c.aux + op.imm + idx * nc_dyn;
        for (u32 j = 0; j < nc_dyn; j++) rowS(c, op.select_row + j, 3, ld_fe(c, tab[j]), index);
        return;
    }
    const u32* tab = c.aux + op.imm + idx * NC;
    Fe v[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) v[j] = ld_fe(c, tab[j]);
#pragma unroll
    for (int j = 0; j < NC; j++) rowS(c, op.select_row + j, 3, v[j], index);
}

template <class FP, bool UNUSED>
WI_INLINE void exec_op(const LC& c, const H2EOp& op);

// Hints are read in a dependent chain; under a saturated memory system such a load costs ~10-100 us.  The ops of
// consecutive ecc ops read the same slots of consecutive 8-slot blocks (tape.h; blocks are 8-aligned), so whenever
// one of the slots the replay needs (lambda^2, t2*lambda, c.x, c.y) is read, the load of the same slot of the
// next block is issued into that slot's register set: it is long finished when the next ecc op asks for it.
template <class FP>
struct HintPrefetch {
    static constexpr int E = 4;
    u32 slot[E];
    Wd<FP::WW> v[E];
};
template <class FP>
WI_INLINE Wd<FP::WW> hint_value(const LC& c, HintPrefetch<FP>& hp, u32 slot) {
    u32 k = slot & (H2E_ECC_HINT_SLOTS - 1);
    int e = k == H2E_HINT_LAMBDA2 ? 0 : k == H2E_HINT_T2L ? 1 : k == H2E_HINT_XC ? 2 : k == H2E_HINT_YC ? 3 : -1;   // wave-uniform
    Wd<FP::WW> r;
    bool hit = false;
#pragma unroll
    for (int i = 0; i < HintPrefetch<FP>::E; i++)
        if (i == e && hp.slot[i] == slot) {
            r = hp.v[i];
            hit = true;
        }
    if (!hit) r = ws_load<FP::WW>(c.hints + (size_t)slot * c.ws);
#pragma unroll
    for (int i = 0; i < HintPrefetch<FP>::E; i++)
        if (i == e) {
            hp.slot[i] = slot + H2E_ECC_HINT_SLOTS;
            hp.v[i] = ws_load<FP::WW>(c.hints + (size_t)(slot + H2E_ECC_HINT_SLOTS) * c.ws);  // workspace has spare slots
        }
    return r;
}
template <class FP, bool UNUSED>
WI_INLINE void exec_op(const LC& c, const H2EOp& op) {
    switch (op.opcode) {
        case H2E_OP_ASSIGN_W: op_assign_w<FP>(c, op); break;
        case H2E_OP_ASSIGN: {
            u32 slot = op.imm + ((op.flags & H2E_FLAG_INPUT_STRIDED) ? c.strand * c.input_stride : 0);
            rowB(c, op.base_row, 1, g_load<4>(c.inputs + (size_t)slot * c.sw), FE0, FE0, FE0, FE0);
        } break;
        case H2E_OP_ASSIGN_BIT: {
            u32 slot = op.imm + ((op.flags & H2E_FLAG_INPUT_STRIDED) ? c.strand * c.input_stride : 0);
            Fe v = g_load<4>(c.inputs + (size_t)slot * c.sw);
            rowB(c, op.base_row, 3, v, v, FE0, FE0, FE0);
        } break;
        case H2E_OP_CONST_INT: op_const_int<FP>(c, op, false); break;
        case H2E_OP_CONST_INT_INPUT: op_const_int<FP>(c, op, true); break;
        case H2E_OP_CONST: rowB(c, op.base_row, 1, g_load<4>(c.pool + op.imm), FE0, FE0, FE0, FE0); break;
        case H2E_OP_INT_ADD: op_int_add<FP>(c, op); break;
        case H2E_OP_INT_SUB: op_int_sub<FP>(c, op); break;
        case H2E_OP_INT_NEG: op_int_neg<FP>(c, op); break;
        case H2E_OP_INT_MUL_SMALL: op_int_mul_small<FP>(c, op); break;
        case H2E_OP_INT_MUL: op_int_mul<FP>(c, op); break;
        case H2E_OP_REDUCE: op_reduce<FP>(c, op); break;
        case H2E_OP_IS_INT_ZERO: op_is_int_zero<FP>(c, op); break;
        case H2E_OP_NOT: {
            Fe x = ld_fe(c, op.refs[0]);
            ROW_B1(c, op.base_row, x, submod_n(c, fe_u64(1), x));
        } break;
        case H2E_OP_MASK_INT: op_mask_int<FP>(c, op); break;
        case H2E_OP_DIV_CORE: op_div_core<FP>(c, op); break;
        case H2E_OP_BISEC_INT: op_bisec_int<FP>(c, op); break;
        case H2E_OP_SUM_LIMBS: op_sum_limbs<FP>(c, op); break;
        case H2E_OP_ASSERT_CONST: op_assert_const(c, op); break;
        case H2E_OP_BISEC: {
            Fe cond = ld_fe(c, op.refs[0]), a = ld_fe(c, op.refs[1]), b = ld_fe(c, op.refs[2]);
            ROW_B4(c, op.base_row, cond, a, cond, b, wd_is_zero<4>(cond) ? b : a);
        } break;
        case H2E_OP_AND: {
            Fe a = ld_fe(c, op.refs[0]), b = ld_fe(c, op.refs[1]);
            ROW_B2(c, op.base_row, a, b, fe_u64(a.v[0] & b.v[0]));
        } break;
        case H2E_OP_OR: {
            Fe a = ld_fe(c, op.refs[0]), b = ld_fe(c, op.refs[1]);
            ROW_B2(c, op.base_row, a, b, fe_u64(a.v[0] | b.v[0]));
        } break;
        case H2E_OP_XNOR: {
            Fe a = ld_fe(c, op.refs[0]), b = ld_fe(c, op.refs[1]);
            ROW_B2(c, op.base_row, a, b, fe_u64(1 ^ a.v[0] ^ b.v[0]));
        } break;
        case H2E_OP_DECOMPOSE_NATIVE: op_decompose_native(c, op); break;
        case H2E_OP_PICK_INDEX: op_pick_index(c, op); break;
        case H2E_OP_CACHE_INT: op_cache_int<FP>(c, op); break;
        case H2E_OP_SELECT_POINT: op_select_point<FP>(c, op); break;
        case H2E_OP_DECOMPOSE_LIMB: op_decompose_limb(c, op); break;
        case H2E_OP_SHIFT_ADD: op_shift_add(c, op); break;
        default: break;
    }
}

// Tape ops are wave-uniform.  A wave pulls 64 ops (4 KB) into LDS with one coalesced load - one HBM/L2 round
// trip per 64 ops instead of one per op, which matters for the latency-bound value chain - and each op is then
// read from LDS at a uniform address and moved to SGPRs, so the opcode switch is a scalar branch and refs / rows
// are scalar operands of the address arithmetic.
#if H2E_COLS_ON
#define H2E_CHUNK_OPS 32u   // (the column unit's LDS is spoken for: 2 KB of ops)
#else
#define H2E_CHUNK_OPS 64u
#endif
struct TapeChunk {
    uint4 w[H2E_CHUNK_OPS][4];  // [op in chunk][4 x 16 bytes]
};
WI_INLINE void load_chunk(TapeChunk* tc, const H2EOp* tape, u32 first, u32 end) {
    u32 lane = threadIdx.x;
    if (lane < H2E_CHUNK_OPS && first + lane < end) {
        const uint4* src = (const uint4*)(tape + first + lane);
#pragma unroll
        for (int k = 0; k < 4; k++) tc->w[lane][k] = src[k];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}
WI_INLINE H2EOp chunk_op(const TapeChunk* tc, u32 k) {
    u32 v[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint4 x = tc->w[k][q];
        v[4 * q + 0] = __builtin_amdgcn_readfirstlane(x.x);
        v[4 * q + 1] = __builtin_amdgcn_readfirstlane(x.y);
        v[4 * q + 2] = __builtin_amdgcn_readfirstlane(x.z);
        v[4 * q + 3] = __builtin_amdgcn_readfirstlane(x.w);
    }
    H2EOp op;
    op.opcode = (uint16_t)(v[0] & 0xffffu);
    op.flags = (uint16_t)(v[0] >> 16);
    op.imm = v[1];
    op.base_row = v[2];
    op.range_row = v[3];
    op.select_row = v[4];
#pragma unroll
    for (int k2 = 0; k2 < H2E_OP_MAX_REFS; k2++) op.refs[k2] = v[5 + k2];
    return op;
}

// Register budgets.  A wave is only dispatched to a SIMD that has its whole register allocation free (512 per lane and
// SIMD).  The expansion keeps every SIMD filled with its own waves; a value-chain kernel of another stream gets a wave in
// when an expansion wave retires *and* what that frees is enough for it - a replay wave with 284 registers never fitted next
// to a remaining 231-register expansion wave and waited for a SIMD to drain completely (6 ms for a 0.5 ms kernel).
#ifndef H2E_X_WAVES
#define H2E_X_WAVES 1
#endif
#ifndef H2E_X_WAVES_WIDE
#define H2E_X_WAVES_WIDE 1   // the same for the 6-word fields (bls12_381: 299 + 43 registers; 2 = capped at 256, 32 of them spilled)
#endif
#ifndef H2E_XP_WAVES
#define H2E_XP_WAVES 2       // the packed expansion: capped at 256 registers (bn256 260 + 4 uncapped: 4 spilled; bls12_381 304 + 48: 50 spilled).
                             // Round 4 measured "1 is better" - under an LDS footprint (41 KB per wave) that let a CU take three waves whatever
                             // the registers allowed; with the op buffer sized by the launch (round 5) a second wave per SIMD fits: 16 x
                             // bls12_381 pipelined 1.199 -> 1.13 ms per step, 8 x bn256 0.587 -> 0.572, 2 x bls12_381 unchanged
#endif
#ifndef H2E_REPLAY_WAVES
#define H2E_REPLAY_WAVES 2
#endif
#ifndef H2E_CHAIN_WAVES
#define H2E_CHAIN_WAVES 1   // predictors / finalize / fix-up kernels: waves per SIMD their register budget must allow (experiments)
#endif
template <class FP, bool VALUES_ONLY>
__global__ void __launch_bounds__(64, (FP::WW > 4 ? H2E_X_WAVES_WIDE : H2E_X_WAVES)) h2e_run_tape(H2ELaunch L, const InstanceDesc* inst, u32 n_instances,
                                                   const H2EFieldConsts* fc) {
    // lanes: [sub-range][strand][instance], each sub-range padded to whole waves so a wave replays one op range; the
    // instance is the minor index: the lanes of a wave are consecutive instances (the minor dimension of the advice
    // arrays), so the wave's loads and stores of a cell are contiguous
    u32 per_sub = n_instances * L.n_strands;
    u32 blocks_per_sub = (per_sub + 63) / 64;
    __shared__ TapeChunk chunk;
    extern __shared__ u64 xcache_dyn[];   // [3][2 L + 4][64] words when the result cache is on (H2ELaunch.rel_refs bit 2), then
    // the stream digest's sums when the run has one: this lane's 3 x 4 sums, [region][word][lane].  Both dynamic: with the chunk's
    // 4 KB a workgroup of a plain run holds 19 KB, so LDS lets a CU take the 8 waves its registers allow (as static arrays: 25 KB, 6)
    u64* dg_sums = xcache_dyn + ((L.rel_refs & 4) ? (size_t)3 * (2 * FP::L + 4) * 64 : 0);
    if (L.rel_refs & 8) __builtin_amdgcn_s_setprio(3);   // the expansion's waves at the chain kernels' priority (g_tune[1] bit 1)
    // One workgroup (= wave) per 64 lanes, or - a launch with fewer workgroups than that (persistent form, L.x_blocks = the
    // real count) - every workgroup takes the blocks blockIdx.x, + gridDim.x, ...: the launch then holds a fixed share of every
    // SIMD's registers for its whole duration and leaves the rest to the kernels of other streams (the next run's value chain),
    // instead of refilling every slot a retiring wave frees before another queue gets a look at it.
    const u32 n_blocks = VALUES_ONLY || L.x_blocks == 0 ? gridDim.x : L.x_blocks;
    for (u32 blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    u32 sub = blk / blocks_per_sub, idx = (blk % blocks_per_sub) * 64 + threadIdx.x;  // sub is wave-uniform
    // padding lanes of the last wave replay the last valid lane's work but store nothing
    bool active = idx < per_sub;
    if (!active) idx = per_sub - 1;
    u32 instance = idx % n_instances, strand = idx / n_instances;
    u32 op_lo = 0, op_hi = L.n_ops;
    if (!VALUES_ONLY && L.n_sub > 1) {
        op_lo = L.sub[sub];
        op_hi = L.sub[sub + 1];
    }
    InstanceDesc d = inst[instance];
    LC c;
    c.base = d.base;
    c.range = d.range;
    c.select = d.select;
    c.inputs = d.inputs;
    c.status = d.status;
    c.ob = L.strand_base0 + strand * L.delta_base;
    c.orr = L.strand_range0 + strand * L.delta_range;
    c.os = L.strand_select0 + strand * L.delta_select;
    c.params = L.params + (size_t)strand * L.n_params;
    c.aux = L.aux;
    c.pool = L.const_pool;
    c.fc = &g_fc[FP::ID];   // constant address space -> scalar loads (the `fc` argument serves the other kernels)
    c.strand = strand;
    c.input_stride = L.input_stride;
    c.sw = L.slot_words;
    c.hints = d.hints;
    c.ws = d.ws;
    c.hint_stride = L.hint_stride;
    c.hs = inst[0].hs;
    c.active = active;
    c.xc = (L.rel_refs & 4) ? xcache_dyn : nullptr;
    if (!VALUES_ONLY && L.dg_out != nullptr) {
#pragma unroll
        for (int k = 0; k < 12; k++) l_st8(dg_sums + k * 64 + threadIdx.x, 0);
        c.dg = dg_sums + threadIdx.x;
    }
    {
        for (u32 i0 = op_lo; i0 < op_hi; i0 += 64) {
            load_chunk(&chunk, L.tape, i0, op_hi);
            u32 n = min(64u, op_hi - i0);
            for (u32 k = 0; k < n; k++) {
                H2EOp op = chunk_op(&chunk, k);
                exec_op<FP, false>(c, op);
            }
        }
    }
    if (c.dg != nullptr && active) {
        lds_fence();
#pragma unroll
        for (int k = 0; k < 12; k++) {
            u64 v = l_ld8(dg_sums + k * 64 + threadIdx.x);
            if (v) atomicAdd((unsigned long long*)(L.dg_out + (((size_t)(blockIdx.x & (L.dg_shards - 1u)) * 3 + k / 4) * n_instances + instance) * 4 + (k % 4)), (unsigned long long)v);
        }
    }
    }
}

#if H2E_COLS_ON
// ------------------------------------------------------------------------------------------------
// The full expansion that stores halo2's advice columns itself (h2e.h h2e_run_columns; the reference's Records::_assign_to_*_chip,
// src/context.rs:310-541, without the pass).  Same lanes, ops and arithmetic as h2e_run_tape<FP, false>; the emission layer above is
// compiled in its staging form.  n_instances is a multiple of 64: a wave = 64 consecutive instances of one strand and sub-range.
template <class FP>
__global__ void __launch_bounds__(64, 1) h2e_run_tape_cols(H2ELaunch L, const InstanceDesc* inst, u32 n_instances) {
    const u32 per_sub = n_instances * L.n_strands, blocks_per_sub = per_sub / 64;
    __shared__ TapeChunk chunk;
    extern __shared__ u64 stg_dyn[];   // base staging [4 slots: columns 0, 1, 2, 4][4][64][4 words], range: [4][64][2 words], 2 x [4][64] dwords
    const u32 blk = blockIdx.x;
    const u32 sub = blk / blocks_per_sub, idx = (blk % blocks_per_sub) * 64 + threadIdx.x;
    const u32 instance = idx % n_instances, strand = idx / n_instances;
    u32 op_lo = 0, op_hi = L.n_ops;
    if (L.n_sub > 1) {
        op_lo = L.sub[sub];
        op_hi = L.sub[sub + 1];
    }
    InstanceDesc d = inst[instance];
    LC c;
    c.base = d.base;
    c.range = d.range;
    c.select = d.select;
    c.inputs = d.inputs;
    c.status = d.status;
    c.ob = L.strand_base0 + strand * L.delta_base;
    c.orr = L.strand_range0 + strand * L.delta_range;
    c.os = L.strand_select0 + strand * L.delta_select;
    c.params = L.params + (size_t)strand * L.n_params;
    c.aux = L.aux;
    c.pool = L.const_pool;
    c.fc = &g_fc[FP::ID];
    c.strand = strand;
    c.input_stride = L.input_stride;
    c.sw = L.slot_words;
    c.hints = d.hints;
    c.ws = d.ws;
    c.hint_stride = L.hint_stride;
    c.hs = inst[0].hs;
    c.active = true;
    c.xc = nullptr;
    c.colB = L.col[0]; c.colR = L.col[1]; c.colS = L.col[2];
    c.csB = L.col_stride[0]; c.csR = L.col_stride[1]; c.csS = L.col_stride[2];
    c.crB = L.col_rows[0]; c.crR = L.col_rows[1]; c.crS = L.col_rows[2];
    c.inst0 = uni(instance - threadIdx.x);
    {
        const u32 lane = threadIdx.x, il = lane >> 3;
        c.fj = (lane & 7u) >> 1;
        c.fhalf = lane & 1u;
#pragma unroll
        for (int s8 = 0; s8 < 8; s8++) {
            c.fB[s8] = L.col[0] + (size_t)(c.inst0 + 8 * s8 + il) * L.col_stride[0] + c.fj * 4 + c.fhalf * 2;
            c.fR[s8] = L.col[1] + (size_t)(c.inst0 + 8 * s8 + il) * L.col_stride[1] + c.fj * 4 + c.fhalf * 2;
        }
    }
    c.stgB = stg_dyn;
    c.stgR0 = stg_dyn + (size_t)4 * 4 * 64 * 4;
    c.stgR12 = (u32*)(c.stgR0 + (size_t)4 * 64 * 2);
    c.lB = c.stgB + ((size_t)c.fj * 64 + (threadIdx.x >> 3)) * 4 + c.fhalf * 2;
    c.lR0 = c.stgR0 + ((size_t)c.fj * 64 + (threadIdx.x >> 3)) * 2;
    c.lR12 = c.stgR12 + (size_t)c.fj * 64 + (threadIdx.x >> 3);
    {
        const H2EOp* first = L.tape + op_lo;
        c.loB = uni(first->base_row + c.ob);
        c.loR = uni(first->range_row + c.orr);
        c.hiB = c.loB;
        c.hiR = c.loR;
    }
    for (u32 i0 = op_lo; i0 < op_hi; i0 += H2E_CHUNK_OPS) {
        load_chunk(&chunk, L.tape, i0, op_hi);
        u32 n = min(H2E_CHUNK_OPS, op_hi - i0);
        for (u32 k = 0; k < n; k++) {
            H2EOp op = chunk_op(&chunk, k);
            exec_op<FP, false>(c, op);
        }
    }
    colB_flush(c, c.hiB);
    colR_flush(c, c.hiR);
}
extern "C" int H2E_UNIT(h2e_engine_set_consts)(int field_pair, const H2EFieldConsts* host) {
    if (field_pair < 0 || field_pair > 2) return -1;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fc), host, sizeof(H2EFieldConsts), (size_t)field_pair * sizeof(H2EFieldConsts),
                                  hipMemcpyHostToDevice);
}
// the expansion of one launch (all its sub-ranges, or sub-ranges [launch->sub ...) of a part) in column-emission form
extern "C" int H2E_UNIT(h2e_engine_launch)(const H2ELaunch* launch, const void* instances, uint32_t n_instances, hipStream_t stream) {
    const u32 per_sub = n_instances * launch->n_strands;
    if (per_sub == 0 || launch->n_ops == 0) return 0;
    if (n_instances % 64 != 0 || !launch->col[0] || !launch->col[1] || !launch->col[2]) return -5;
    const u32 n_sub = launch->n_sub > 1 ? launch->n_sub : 1;
    const size_t lds = ((size_t)4 * 4 * 64 * 4 + (size_t)4 * 64 * 2) * 8 + (size_t)2 * 4 * 64 * 4;   // 32 KB + 4 KB + 2 KB (+ 2 KB of ops): four waves per CU
    H2ELaunch l = *launch;
    l.rel_refs &= ~4u;
#if H2E_FP_ONLY == 0
    hipLaunchKernelGGL(h2e_run_tape_cols<FP_BN256_FQ>, dim3(per_sub / 64 * n_sub), dim3(64), lds, stream, l, (const InstanceDesc*)instances, n_instances);
#elif H2E_FP_ONLY == 1
    hipLaunchKernelGGL(h2e_run_tape_cols<FP_BLS_FQ>, dim3(per_sub / 64 * n_sub), dim3(64), lds, stream, l, (const InstanceDesc*)instances, n_instances);
#else
    hipLaunchKernelGGL(h2e_run_tape_cols<FP_BLS_FR>, dim3(per_sub / 64 * n_sub), dim3(64), lds, stream, l, (const InstanceDesc*)instances, n_instances);
#endif
    return (int)hipGetLastError();
}
#else   // everything below: the plain units only

// ------------------------------------------------------------------------------------------------
// Packed expansion: batches smaller than a wave.  h2e_run_tape's lanes are (strand, instance) of ONE sub-range, so a launch
// with n_strands x n_instances < 64 - a pairing check has one strand: 16 bls12_381 checks per GPU (configs[4]), 8 bn256 / 2
// bls12_381 checks when the batch is dealt over 8 GPUs - leaves most of every wave instruction idle and stores 256 / 128 / 32-byte
// runs.  Here a wave takes G = 64 / P sub-ranges at once (P = the lanes of one sub-range rounded up to a power of two): lane
// = (group, instance).  The groups run different op sequences, so an op is per-lane data (rows and cell references in VGPRs)
// and the wave executes one opcode at a time: every group looks at the op at its own cursor, the wave picks the LIGHTEST
// opcode any cursor shows and runs it for the groups that show it.  Light ops (additions, selections, conditions) so run ahead
// until every cursor stands at a multiplication-like op (int_mul / div / reduce: 3/4 of the cells), and those run for all groups
// in one pass - the wave synchronises itself on the heavy ops without any host-side alignment of the sub-ranges.
// Op records reach the lanes through LDS: each group keeps a chunk of min(32, 256 / G) ops - a whole sub-range of the pairing programs -
// loaded by the group's own lanes.  (A refill waits for the loads, and with them for every store the wave has in flight - one counter, in
// order: with 8-op chunks and 8 groups that was a store drain in nearly every step, 13 us per step against 8 us per op unpacked.)
#define H2E_PK_BUF_OPS 256u   // ops the kernel's LDS buffer holds (16 KB), shared by the wave's groups
// opcodes from light to heavy (cells an op writes ~ the instructions it costs)
static __constant__ unsigned char g_pk_rank_of[H2E_OP_COUNT] = {
    /* NOP */ 0, /* ASSIGN_W */ 20, /* ASSIGN */ 1, /* ASSIGN_BIT */ 2, /* CONST_INT */ 12, /* CONST_INT_INPUT */ 13, /* CONST */ 3,
    /* INT_ADD */ 14, /* INT_SUB */ 15, /* INT_NEG */ 16, /* INT_MUL_SMALL */ 17, /* INT_MUL */ 30, /* REDUCE */ 28, /* IS_INT_ZERO */ 22,
    /* NOT */ 4, /* MASK_INT */ 18, /* DIV_CORE */ 31, /* BISEC_INT */ 19, /* SUM_LIMBS */ 5, /* ASSERT_CONST */ 6, /* BISEC */ 7, /* AND */ 8,
    /* OR */ 9, /* XNOR */ 10, /* DECOMPOSE_NATIVE */ 26, /* PICK_INDEX */ 11, /* CACHE_INT */ 21, /* SELECT_POINT */ 23,
    /* DECOMPOSE_LIMB */ 25, /* SHIFT_ADD */ 24};
static_assert(H2E_OP_COUNT == 30, "g_pk_rank_of lists every opcode of tape.h");
template <class FP>
__global__ void __launch_bounds__(64, H2E_XP_WAVES) h2e_run_tape_packed(H2ELaunch L, const InstanceDesc* inst, u32 n_instances, u32 log2p) {
    __shared__ u32 rank_lds[32];                  // g_pk_rank_of, read per lane in the loop: from LDS (a global load there would wait
                                                  // for every store the wave has in flight - one counter, in order)
    // dynamic LDS, sized by the launch (pk_lds_bytes below): the op buffer [group][op in chunk][4 x 16 bytes] - G x CH ops: 16 KB for eight
    // groups and more, 8 KB for four -, the digest sums (6 KB, only a run with a stream digest), the result cache [3][2 L + 4][64] words
    // (H2ELaunch.rel_refs bit 2).  With 16 KB + 6 KB static whatever the launch needed, a wave of 16 bls12_381 checks held 41 KB and a CU
    // took THREE of them - one of its four SIMDs idle, and none next to a chain's workgroup.
    extern __shared__ u64 pk_dyn[];
    if (L.rel_refs & 8) __builtin_amdgcn_s_setprio(3);
    const u32 lane = threadIdx.x, P = 1u << log2p, G = 64u >> log2p;
    if (lane < 32u) rank_lds[lane] = lane < (u32)H2E_OP_COUNT ? (u32)g_pk_rank_of[lane] : 0xffu;
    const u32 g = lane >> log2p, ii = lane & (P - 1u);
    const u32 CH = min(32u, H2E_PK_BUF_OPS / G);   // ops per group and chunk (sub-ranges of the pairing programs: 8 or 16-17 ops)
    u32x4* opbuf = (u32x4*)pk_dyn;
    u64* dg_sums = pk_dyn + (size_t)G * CH * 8u;
    u64* xcache_dyn = dg_sums + (L.dg_out != nullptr ? 12u * 64u : 0u);
    const u32 per_sub = n_instances * L.n_strands;
    const u32 n_sub = L.n_sub > 1 ? L.n_sub : 1;
    const u32 sub = L.pk_order ? L.pk_order[blockIdx.x * G + g] : blockIdx.x * G + g;   // (~0u: an empty slot of the order table)
    const bool group_on = sub < n_sub;
    const u32 idx = ii < per_sub ? ii : per_sub - 1;      // padding lanes of a group replay its last lane and store nothing
    const u32 instance = idx % n_instances, strand = idx / n_instances;
    u32 pos = 0, end = 0;
    if (group_on) {
        pos = L.n_sub > 1 ? L.sub[sub] : 0u;
        end = L.n_sub > 1 ? L.sub[sub + 1] : L.n_ops;
    }
    InstanceDesc d = inst[instance];
    LC c;
    c.base = d.base;
    c.range = d.range;
    c.select = d.select;
    c.inputs = d.inputs;
    c.status = d.status;
    c.ob = L.strand_base0 + strand * L.delta_base;
    c.orr = L.strand_range0 + strand * L.delta_range;
    c.os = L.strand_select0 + strand * L.delta_select;
    c.params = L.params + (size_t)strand * L.n_params;
    c.aux = L.aux;
    c.pool = L.const_pool;
    c.fc = &g_fc[FP::ID];
    c.strand = strand;
    c.input_stride = L.input_stride;
    c.sw = L.slot_words;
    c.hints = d.hints;
    c.ws = d.ws;
    c.hint_stride = L.hint_stride;
    c.hs = inst[0].hs;
    c.active = group_on && ii < per_sub;
    c.xc = (L.rel_refs & 4) ? xcache_dyn : nullptr;
    if (L.dg_out != nullptr) {
#pragma unroll
        for (int k = 0; k < 12; k++) l_st8(dg_sums + k * 64 + lane, 0);
        c.dg = dg_sums + lane;
    }
    u32 cbase = pos;          // first op of the chunk this group holds in LDS ...
    bool filled = false;      // ... once it has loaded one
    const H2E_AS_LDS u32x4* my_ops = (const H2E_AS_LDS u32x4*)opbuf + (size_t)g * CH * 4;
    for (;;) {
        const bool need = pos < end && (!filled || pos >= cbase + CH);
        if (__builtin_amdgcn_ballot_w64(need)) {
            if (need) {
                cbase = pos;
                filled = true;
                for (u32 e = ii; e < CH * 4u; e += P) {
                    u32 at = cbase + e / 4u;
                    if (at < end) ((H2E_AS_LDS u32x4*)opbuf)[((size_t)g * CH + e / 4u) * 4u + e % 4u] =
                        ((const H2E_AS_GLOBAL u32x4*)(L.tape + at))[e % 4u];
                }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        u32 rank = 0xffu;
        if (pos < end) {
            u32 w0 = my_ops[(pos - cbase) * 4u].x;
            rank = ((const H2E_AS_LDS u32*)rank_lds)[w0 & 31u];
        }
        // the lightest op any cursor shows (ranks are below 32)
        u32 pick = 0xffu;
        for (u32 r = 0; r < 32u; r++)
            if (__builtin_amdgcn_ballot_w64(rank == r)) {
                pick = r;
                break;
            }
        if (pick == 0xffu) break;   // every group is through its sub-range
        if (rank == pick) {
            const H2E_AS_LDS u32x4* q = my_ops + (pos - cbase) * 4u;
            u32x4 x0 = q[0], x1 = q[1], x2 = q[2], x3 = q[3];
            H2EOp op;
            op.opcode = (uint16_t)__builtin_amdgcn_readfirstlane(x0.x & 0xffffu);   // the same for every lane in here: a scalar switch
            op.flags = (uint16_t)(x0.x >> 16);
            op.imm = x0.y;
            op.base_row = x0.z;
            op.range_row = x0.w;
            op.select_row = x1.x;
            op.refs[0] = x1.y; op.refs[1] = x1.z; op.refs[2] = x1.w;
            op.refs[3] = x2.x; op.refs[4] = x2.y; op.refs[5] = x2.z; op.refs[6] = x2.w;
            op.refs[7] = x3.x; op.refs[8] = x3.y; op.refs[9] = x3.z; op.refs[10] = x3.w;
            exec_op<FP, false>(c, op);
            pos++;
        }
    }
    if (c.dg != nullptr && c.active) {
        lds_fence();
#pragma unroll
        for (int k = 0; k < 12; k++) {
            u64 v = l_ld8(dg_sums + k * 64 + lane);
            if (v) atomicAdd((unsigned long long*)(L.dg_out + (((size_t)(blockIdx.x & (L.dg_shards - 1u)) * 3 + k / 4) * n_instances + instance) * 4 + (k % 4)), (unsigned long long)v);
        }
    }
}

// ================================================================================================
// Compiled values-only replay (tape.h "V-tape"): the critical path of every cut segment.  One wave walks the
// record stream; operands and results live in statically allocated LDS slots ([slot][word][lane], conflict-free),
// so an op is: read its header (LDS, uniform address -> SGPRs), read operands (one LDS round trip), compute,
// write the slot and - only where the host's liveness analysis says someone else needs it - the cells.
template <class FP>
struct VSlots {   // views into the kernel's dynamic LDS
    static constexpr int W = 2 * FP::L + 4, NF = 4;
    static constexpr int INT_UNITS = FP::L + 2, HINT_UNITS = FP::WW / 2;   // 16-byte staging units of an integer / a hint
    u64* ints;          // [n_int_slots][W][64]
    u64* fes;           // [NF][4][64]
    ulonglong2* stage;  // [n_units][64]
};
template <class FP>
WI_INLINE IntVal<FP> vs_ld_int(const VSlots<FP>& vs, u32 slot) {
    IntVal<FP> r;
    const u64* p = vs.ints + (size_t)slot * VSlots<FP>::W * 64 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < FP::L; i++) {
        r.l[i].v[0] = l_ld8(p + (2 * i) * 64);
        r.l[i].v[1] = l_ld8(p + (2 * i + 1) * 64);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) r.native.v[i] = l_ld8(p + (2 * FP::L + i) * 64);
    return r;
}
template <class FP>
WI_INLINE void vs_st_int(const VSlots<FP>& vs, u32 slot, const Limb* l, const Fe& native) {
    u64* p = vs.ints + (size_t)slot * VSlots<FP>::W * 64 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < FP::L; i++) {
        l_st8(p + (2 * i) * 64, l[i].v[0]);
        l_st8(p + (2 * i + 1) * 64, l[i].v[1]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) l_st8(p + (2 * FP::L + i) * 64, native.v[i]);
}
template <class FP>
WI_INLINE Fe vs_ld_fe(const VSlots<FP>& vs, u32 slot) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.v[i] = l_ld8(vs.fes + (slot * 4 + i) * 64 + threadIdx.x);
    return r;
}
template <class FP>
WI_INLINE void vs_st_fe(const VSlots<FP>& vs, u32 slot, const Fe& v) {
#pragma unroll
    for (int i = 0; i < 4; i++) l_st8(vs.fes + (slot * 4 + i) * 64 + threadIdx.x, v.v[i]);
}
template <class FP>
WI_INLINE IntVal<FP> vs_stage_int(const VSlots<FP>& vs, u32 unit) {   // L limb units + 2 native units
    IntVal<FP> r;
#pragma unroll
    for (int i = 0; i < FP::L; i++) {
        u64x2 q = l_ld16((const u64*)(vs.stage + (unit + i) * 64 + threadIdx.x));
        r.l[i].v[0] = q.x;
        r.l[i].v[1] = q.y;
    }
    u64x2 a = l_ld16((const u64*)(vs.stage + (unit + FP::L) * 64 + threadIdx.x)), b = l_ld16((const u64*)(vs.stage + (unit + FP::L + 1) * 64 + threadIdx.x));
    r.native.v[0] = a.x; r.native.v[1] = a.y; r.native.v[2] = b.x; r.native.v[3] = b.y;
    return r;
}
template <class FP>
WI_INLINE Fe vs_stage_fe(const VSlots<FP>& vs, u32 unit) {
    u64x2 a = l_ld16((const u64*)(vs.stage + unit * 64 + threadIdx.x)), b = l_ld16((const u64*)(vs.stage + (unit + 1) * 64 + threadIdx.x));
    Fe r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b.x; r.v[3] = b.y;
    return r;
}
template <class FP>
WI_INLINE Wd<FP::WW> vs_stage_w(const VSlots<FP>& vs, u32 unit) {
    Wd<FP::WW> r;
#pragma unroll
    for (int i = 0; i < FP::WW / 2; i++) {
        u64x2 q = l_ld16((const u64*)(vs.stage + (unit + i) * 64 + threadIdx.x));
        r.v[2 * i] = q.x;
        r.v[2 * i + 1] = q.y;
    }
    return r;
}
struct VHdr {
    u32 w[8];
};
WI_INLINE VHdr vrec_read(const H2EVRec* rec) {  // uniform address: every lane reads the same record
    const uint4* q = (const uint4*)rec;
    uint4 a = q[0], b = q[1];
    VHdr h;
    h.w[0] = __builtin_amdgcn_readfirstlane(a.x);
    h.w[1] = __builtin_amdgcn_readfirstlane(a.y);
    h.w[2] = __builtin_amdgcn_readfirstlane(a.z);
    h.w[3] = __builtin_amdgcn_readfirstlane(a.w);
    h.w[4] = __builtin_amdgcn_readfirstlane(b.x);
    h.w[5] = __builtin_amdgcn_readfirstlane(b.y);
    h.w[6] = __builtin_amdgcn_readfirstlane(b.z);
    h.w[7] = __builtin_amdgcn_readfirstlane(b.w);
    return h;
}
template <class FP>
WI_INLINE IntVal<FP> v_src_int(const VSlots<FP>& vs, const LC& c, const VHdr& h, int which, const H2EVRec* ext) {
    u32 kind = (h.w[7] >> (3 * which)) & 7u, word = h.w[2 + which];
    if (kind == H2E_VSRC_INT_SLOT) return vs_ld_int<FP>(vs, word);
    if (kind == H2E_VSRC_STAGE) return vs_stage_int<FP>(vs, word);
    u32 refs[FP::L + 1];
    const u32* e = (const u32*)ext + word;
#pragma unroll
    for (int j = 0; j <= FP::L; j++) refs[j] = __builtin_amdgcn_readfirstlane(e[j]);
    return ld_int<FP>(c, refs);
}
template <class FP>
WI_INLINE Fe v_src_fe(const VSlots<FP>& vs, const LC& c, const VHdr& h, int which) {
    u32 kind = (h.w[7] >> (3 * which)) & 7u, word = h.w[2 + which];
    if (kind == H2E_VSRC_FE_SLOT) return vs_ld_fe<FP>(vs, word);
    if (kind == H2E_VSRC_STAGE) return vs_stage_fe<FP>(vs, word);
    return ld_fe(c, word);
}
// result cells: mul-like = limbs in range acc cells + native in a base cell; add-like = base column 4
template <class FP>
WI_INLINE void v_out_mul(const VSlots<FP>& vs, const LC& c, const VHdr& h, const Limb* l, const Fe& native) {
    if ((h.w[0] >> 8) & H2E_VFLAG_STORE) {
#pragma unroll
        for (int i = 0; i < FP::L; i++) stR(c, h.w[6] + 3 * i, 0, fe_of(l[i]));
        stB(c, h.w[5], 4, native);
    }
    u32 dst = (h.w[0] >> 16) & 0xffu;
    if (dst != H2E_V_NO_SLOT) vs_st_int<FP>(vs, dst, l, native);
}
template <class FP>
WI_INLINE void v_out_add(const VSlots<FP>& vs, const LC& c, const VHdr& h, const Limb* l, const Fe& native) {
    if ((h.w[0] >> 8) & H2E_VFLAG_STORE) {
#pragma unroll
        for (int i = 0; i < FP::L; i++) stB(c, h.w[5] + i, 4, fe_of(l[i]));
        stB(c, h.w[5] + FP::L, 4, native);
    }
    u32 dst = (h.w[0] >> 16) & 0xffu;
    if (dst != H2E_V_NO_SLOT) vs_st_int<FP>(vs, dst, l, native);
}
template <class FP>
WI_INLINE void v_out_fe(const VSlots<FP>& vs, const LC& c, const VHdr& h, const Fe& v) {
    if ((h.w[0] >> 8) & H2E_VFLAG_STORE) stB(c, h.w[5], 4, v);
    u32 dst = (h.w[0] >> 16) & 0xffu;
    if (dst != H2E_V_NO_SLOT) vs_st_fe<FP>(vs, dst, v);
}
template <class FP>
WI_INLINE void v_out_w(const VSlots<FP>& vs, const LC& c, const VHdr& h, const Wd<FP::WW>& x) {  // canonical W value
    Limb l[FP::L];
    split_limbs<FP>(x, l);
    v_out_mul<FP>(vs, c, h, l, native_of_w<FP>(c, x));
}

template <class FP>
WI_INLINE void exec_vop(const VSlots<FP>& vs, const LC& c, const VHdr& h, const H2EVRec* ext, HintPrefetch<FP>& hp) {
    constexpr int L = FP::L;
    u32 opc = h.w[0] & 0xffu, imm = h.w[1];
    if (opc == H2E_V_HINT) {
        if ((h.w[0] >> 8) & H2E_VFLAG_STAGED) {
            v_out_w<FP>(vs, c, h, vs_stage_w<FP>(vs, imm));
        } else {
            u32 slot = imm + ((((h.w[0] >> 8) & H2E_VFLAG_HINT_STRIDED) != 0) ? c.strand * c.hint_stride : 0);
            v_out_w<FP>(vs, c, h, hint_value<FP>(c, hp, slot));
        }
        return;
    }
    if (opc == H2E_V_GATHER) {   // asynchronous global -> LDS loads of this piece's inputs (16 bytes per lane and unit)
        u32 n = (h.w[0] >> 8) & 0xffu;
#pragma unroll
        for (u32 e = 0; e < 3; e++)
            if (e < n) {
                u32 meta = h.w[2 + 2 * e], ref = h.w[3 + 2 * e];
                const u64* src;
                u32 piece = (meta >> 4) & 0xfu;   // 16-byte piece of the value
                if ((meta & 3u) == 1u)
                    src = c.hints + (size_t)(ref + ((meta & 0x100u) ? c.strand * c.hint_stride : 0)) * c.ws + piece * 2u;
                else if ((meta & 3u) == 2u)   // x then y: two value slots, H2E_W_WORDS_MAX / 2 piece numbers each
                    src = c.sel + (size_t)(H2E_SEL_SLOTS * (ref + c.strand * c.sel_stride) + piece / (H2E_W_WORDS_MAX / 2)) * c.ws + (piece % (H2E_W_WORDS_MAX / 2)) * 2u;
                else
                    src = cell_ptr(c, ref) + (size_t)piece * c.hs;   // a cell's halves are hs words apart
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(vs.stage + (size_t)(h.w[1] + e) * 64), 16, 0, 0);
            }
        return;
    }
    if (opc == H2E_V_GATHER_WAIT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    if (opc == H2E_V_SUB) {
        IntVal<FP> a = v_src_int<FP>(vs, c, h, 0, ext), b = v_src_int<FP>(vs, c, h, 1, ext);
        Limb s[L];
#pragma unroll
        for (int i = 0; i < L; i++) s[i] = wd_sub<2>(wd_add<2>(a.l[i], wd_load<2>(c.fc->ceil_limbs[imm][i])), b.l[i]);
        v_out_add<FP>(vs, c, h, s, addmod_n(c, submod_n(c, a.native, b.native), wd_load<4>(c.fc->ceil_native[imm])));
        return;
    }
    if (opc == H2E_V_ADD) {
        IntVal<FP> a = v_src_int<FP>(vs, c, h, 0, ext), b = v_src_int<FP>(vs, c, h, 1, ext);
        Limb s[L];
#pragma unroll
        for (int i = 0; i < L; i++) s[i] = wd_add<2>(a.l[i], b.l[i]);
        v_out_add<FP>(vs, c, h, s, addmod_n(c, a.native, b.native));
        return;
    }
    if (opc == H2E_V_MUL) {
        IntVal<FP> a = v_src_int<FP>(vs, c, h, 0, ext), b = v_src_int<FP>(vs, c, h, 1, ext);
        Wd<FPX<FP>::AW> A = compose<FP, FPX<FP>::AW>(a.l), B = compose<FP, FPX<FP>::AW>(b.l);
        Wd<FPX<FP>::QW> dq;
        Wd<FP::WW> rem;
        divrem_w<FP>(c, wd_resize<FPX<FP>::XW>(wd_mul<FPX<FP>::AW, FPX<FP>::AW, FPX<FP>::AL, FPX<FP>::AL>(A, B)), dq, rem);
        v_out_w<FP>(vs, c, h, rem);
        return;
    }
    if (opc == H2E_V_REDUCE) {
        IntVal<FP> a = v_src_int<FP>(vs, c, h, 0, ext);
        Wd<FP::WW> rem;
        u64 dsmall;
        divrem_small<FP>(c, compose<FP, FPX<FP>::AW>(a.l), dsmall, rem);
        v_out_w<FP>(vs, c, h, rem);
        return;
    }
    if (opc == H2E_V_NEG) {
        IntVal<FP> a = v_src_int<FP>(vs, c, h, 0, ext);
        Limb s[L];
#pragma unroll
        for (int i = 0; i < L; i++) s[i] = wd_sub<2>(wd_load<2>(c.fc->ceil_limbs[imm][i]), a.l[i]);
        v_out_add<FP>(vs, c, h, s, submod_n(c, wd_load<4>(c.fc->ceil_native[imm]), a.native));
        return;
    }
    if (opc == H2E_V_MUL_SMALL) {
        IntVal<FP> a = v_src_int<FP>(vs, c, h, 0, ext);
        Wd<1> k = wd_from_u64<1>(imm);
        Limb s[L];
#pragma unroll
        for (int i = 0; i < L; i++) s[i] = wd_resize<2>(wd_mul<2, 1>(a.l[i], k));
        v_out_add<FP>(vs, c, h, s, mod_n<5>(c, wd_mul<4, 1>(a.native, k)));
        return;
    }
    if (opc == H2E_V_DIV) {  // unhinted division: b, a
        IntVal<FP> b = v_src_int<FP>(vs, c, h, 0, ext), a = v_src_int<FP>(vs, c, h, 1, ext);
        Wd<FPX<FP>::QW> q0;
        Wd<FP::WW> a_red, b_red, cv;
        divrem_w<FP>(c, wd_resize<FPX<FP>::XW>(compose<FP, FPX<FP>::AW>(a.l)), q0, a_red);
        divrem_w<FP>(c, wd_resize<FPX<FP>::XW>(compose<FP, FPX<FP>::AW>(b.l)), q0, b_red);
        Wd<FP::WW> binv = wd_inv_mod<FP::WW>(b_red, wd_load<FP::WW>(c.fc->w));
        divrem_w<FP>(c, wd_resize<FPX<FP>::XW>(wd_mul<FP::WW, FP::WW>(a_red, binv)), q0, cv);
        v_out_w<FP>(vs, c, h, cv);
        return;
    }
    if (opc == H2E_V_MASK) {
        IntVal<FP> a = v_src_int<FP>(vs, c, h, 0, ext);
        Fe coeff = v_src_fe<FP>(vs, c, h, 1);
        bool keep = !wd_is_zero<4>(coeff);
        Limb s[L];
#pragma unroll
        for (int i = 0; i < L; i++) s[i] = keep ? a.l[i] : wd_zero<2>();
        v_out_add<FP>(vs, c, h, s, keep ? a.native : wd_zero<4>());
        return;
    }
    if (opc == H2E_V_BISEC_INT) {
        Fe cond = v_src_fe<FP>(vs, c, h, 0);
        IntVal<FP> a = v_src_int<FP>(vs, c, h, 1, ext), b = v_src_int<FP>(vs, c, h, 2, ext);
        bool take_a = !wd_is_zero<4>(cond);
        Limb s[L];
#pragma unroll
        for (int i = 0; i < L; i++) s[i] = take_a ? a.l[i] : b.l[i];
        v_out_add<FP>(vs, c, h, s, take_a ? a.native : b.native);
        return;
    }
    if (opc == H2E_V_IS_ZERO) {
        IntVal<FP> a = v_src_int<FP>(vs, c, h, 0, ext);
        bool all_zero = true;
#pragma unroll
        for (int i = 0; i < L; i++) all_zero = all_zero && wd_is_zero<2>(a.l[i]);
        bool is_w = wd_eq<4>(a.native, wd_load<4>(c.fc->w_native));
#pragma unroll
        for (int i = 0; i < FP::PW; i++) is_w = is_w && wd_eq<2>(a.l[i], wd_load<2>(c.fc->w_limbs[i]));
        v_out_fe<FP>(vs, c, h, fe_u64((all_zero || is_w) ? 1 : 0));
        return;
    }
    if (opc == H2E_V_NOT) {
        v_out_fe<FP>(vs, c, h, submod_n(c, fe_u64(1), v_src_fe<FP>(vs, c, h, 0)));
        return;
    }
    if (opc == H2E_V_AND || opc == H2E_V_OR || opc == H2E_V_XNOR) {
        Fe a = v_src_fe<FP>(vs, c, h, 0), b = v_src_fe<FP>(vs, c, h, 1);
        u64 r = opc == H2E_V_AND ? (a.v[0] & b.v[0]) : opc == H2E_V_OR ? (a.v[0] | b.v[0]) : (1 ^ a.v[0] ^ b.v[0]);
        v_out_fe<FP>(vs, c, h, fe_u64(r));
        return;
    }
    if (opc == H2E_V_PICK_INDEX) {  // imm = k, the bit cells' refs are the first k extension words
        u64 idx = 0;
        const u32* e = (const u32*)ext;
#pragma unroll
        for (int i = 0; i < 5; i++)
            if (i < (int)imm) idx |= (ld_fe(c, __builtin_amdgcn_readfirstlane(e[i])).v[0] & 1) << i;
        v_out_fe<FP>(vs, c, h, fe_u64(idx));
        return;
    }
    if (opc == H2E_V_SELECT_POINT) {  // src0 = index, imm = aux offset of the candidate table, w[6] = select row
        constexpr int NC = 2 * (L + 1);
        Fe index = v_src_fe<FP>(vs, c, h, 0);
        u32 idx = (u32)(index.v[0] & 0xff);
        const u32* tab = c.aux + imm + idx * NC;
        Fe v[NC];
#pragma unroll
        for (int j = 0; j < NC; j++) v[j] = ld_fe(c, tab[j]);
        if ((h.w[0] >> 8) & H2E_VFLAG_STORE) {
#pragma unroll
            for (int j = 0; j < NC; j++) stS(c, h.w[6] + j, 0, v[j]);
        }
        u32 dst[2] = {(h.w[0] >> 16) & 0xffu, (h.w[7] >> 16) & 0xffu};
#pragma unroll
        for (int which = 0; which < 2; which++) {
            Limb l[L];
#pragma unroll
            for (int i = 0; i < L; i++) l[i] = wd_resize<2>(v[which * (L + 1) + i]);
            if (dst[which] != H2E_V_NO_SLOT) vs_st_int<FP>(vs, dst[which], l, v[which * (L + 1) + L]);
        }
        return;
    }
    if (opc == H2E_V_CONST) {   // a pool constant: limb i in (base_row + i, col 0), native in (base_row + L, col 0)
        Wd<FP::WW> x = g_load<FP::WW>(c.pool + imm);
        Limb l[L];
        split_limbs<FP>(x, l);
        Fe native = mod_n<FP::WW>(c, x);
        if ((h.w[0] >> 8) & H2E_VFLAG_STORE) {
#pragma unroll
            for (int i = 0; i < L; i++) stB(c, h.w[5] + i, 0, fe_of(l[i]));
            stB(c, h.w[5] + L, 0, native);
        }
        u32 dst = (h.w[0] >> 16) & 0xffu;
        if (dst != H2E_V_NO_SLOT) vs_st_int<FP>(vs, dst, l, native);
        return;
    }
    if (opc == H2E_V_LOAD_SEL) {   // a point picked by the select pre-kernel: x, y canonical -> two integer slots
        Wd<FP::WW> xy[2];
        if (((h.w[7]) & 7u) == H2E_VSRC_STAGE) {
            xy[0] = vs_stage_w<FP>(vs, h.w[2]);
            xy[1] = vs_stage_w<FP>(vs, h.w[2] + VSlots<FP>::HINT_UNITS);
        } else {
            const u64* p = c.sel + (size_t)H2E_SEL_SLOTS * (h.w[2] + c.strand * c.sel_stride) * c.ws;
            xy[0] = ws_load<FP::WW>(p);
            xy[1] = ws_load<FP::WW>(p + c.ws);
        }
        u32 dst[2] = {(h.w[0] >> 16) & 0xffu, (h.w[7] >> 16) & 0xffu};
#pragma unroll
        for (int which = 0; which < 2; which++) {
            Limb l[L];
            split_limbs<FP>(xy[which], l);
            if (dst[which] != H2E_V_NO_SLOT) vs_st_int<FP>(vs, dst[which], l, native_of_w<FP>(c, xy[which]));
        }
        return;
    }
    if (opc == H2E_V_FULL) {  // the tape op itself (64 bytes = two extension records)
        const u32* e = (const u32*)ext;
        u32 v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = __builtin_amdgcn_readfirstlane(e[q]);
        H2EOp op;
        op.opcode = (uint16_t)(v[0] & 0xffffu);
        op.flags = (uint16_t)(v[0] >> 16);
        op.imm = v[1];
        op.base_row = v[2];
        op.range_row = v[3];
        op.select_row = v[4];
#pragma unroll
        for (int k2 = 0; k2 < H2E_OP_MAX_REFS; k2++) op.refs[k2] = v[5 + k2];
        exec_op<FP, false>(c, op);
        return;
    }
}

// (a six-word field does not fit the two-waves-per-SIMD register budget: 100 spilled registers for bls12_381 Fq, whose replays -
// the general-scalar MSM's candidates - are small launches anyway: one wave per SIMD there)
template <class FP>
__global__ void __launch_bounds__(64, (FP::WW > 4 ? 1 : H2E_REPLAY_WAVES)) h2e_replay(H2ELaunch L, const InstanceDesc* inst, u32 n_instances) {
    // lanes: [piece][instance][strand]; pieces are independent stretches of the replay (host: compile_replay)
    u32 per = n_instances * L.n_strands;
    u32 blocks_per = (per + 63) / 64;
    u32 piece = blockIdx.x / blocks_per;
    u32 idx = (blockIdx.x % blocks_per) * 64 + threadIdx.x;
    bool active = idx < per;
    if (!active) idx = per - 1;   // padding lanes repeat the last lane's work (same values to the same cells)
    u32 rec0 = L.vpieces[2 * piece], rec1 = L.vpieces[2 * piece + 1];
    u32 instance = idx % n_instances, strand = idx / n_instances;   // instance minor, like the advice arrays
    InstanceDesc d = inst[instance];
    LC c;
    c.base = d.base;
    c.range = d.range;
    c.select = d.select;
    c.inputs = d.inputs;
    c.status = d.status;
    c.ob = L.strand_base0 + strand * L.delta_base;
    c.orr = L.strand_range0 + strand * L.delta_range;
    c.os = L.strand_select0 + strand * L.delta_select;
    c.params = L.params + (size_t)strand * L.n_params;
    c.aux = L.aux;
    c.pool = L.const_pool;
    c.fc = &g_fc[FP::ID];
    c.strand = strand;
    c.input_stride = L.input_stride;
    c.sw = L.slot_words;
    c.hints = d.hints;
    c.ws = d.ws;
    c.hint_stride = L.hint_stride;
    c.sel = d.sel;
    c.sel_stride = L.sel_stride;
    c.hs = inst[0].hs;
    __shared__ H2EVRec chunk[2][H2E_VCHUNK];
    extern __shared__ ulonglong2 v_dyn[];
    VSlots<FP> slots;
    slots.stage = v_dyn;
    slots.ints = (u64*)(v_dyn + (size_t)L.v_units * 64);
    slots.fes = slots.ints + (size_t)L.v_int_slots * VSlots<FP>::W * 64;
    c.active = active;
    __builtin_amdgcn_s_setprio(3);   // the value chain is the critical path
    HintPrefetch<FP> hp;
#pragma unroll
    for (int i = 0; i < HintPrefetch<FP>::E; i++) {
        hp.slot[i] = 0xffffffffu;
        hp.v[i] = wd_zero<FP::WW>();
    }
    // records stream through two LDS chunk buffers: the next chunk's global loads are in flight while this one runs
    u32 lane = threadIdx.x;
    auto fetch = [&](u32 first, uint4* r) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            u32 rec = first + 2 * lane + (u32)(q >> 1);
            r[q] = rec < rec1 ? ((const uint4*)(L.vtape + rec))[q & 1] : make_uint4(0, 0, 0, 0);
        }
    };
    auto commit = [&](u32 buf, const uint4* r) {
#pragma unroll
        for (int q = 0; q < 4; q++) ((uint4*)&chunk[buf][2 * lane + (q >> 1)])[q & 1] = r[q];
        lds_fence();
    };
    uint4 nxt[4];
    fetch(rec0, nxt);
    commit(0, nxt);
    for (u32 c0 = rec0, buf = 0; c0 < rec1; c0 += H2E_VCHUNK, buf ^= 1) {
        bool more = c0 + H2E_VCHUNK < rec1;
        if (more) fetch(c0 + H2E_VCHUNK, nxt);
        u32 n = min(H2E_VCHUNK, rec1 - c0);
        for (u32 k = 0; k < n;) {
            VHdr h = vrec_read(&chunk[buf][k]);
            exec_vop<FP>(slots, c, h, &chunk[buf][k + 1], hp);
            k += 1 + (h.w[0] >> 24);
        }
        if (more) commit(buf ^ 1, nxt);
    }
}

// ================================================================================================
// Level-parallel replay (host: compile_replay in h2e_capi.cpp).  A wave replays ONE (instance, strand); lanes are given
// to independent ops: step s runs records 64 s .. 64 s + 63, all of one opcode (H2E_V_NOP in unused lanes), values live
// in LDS slots shared by the wave.  For a pairing check this is ~30 k steps instead of a chain of 175 k ops.
template <class FP>
struct LVals {
    static constexpr int W = 2 * FP::L + 4;
    u64* v;   // [slot][W]
};
template <class FP>
WI_INLINE IntVal<FP> lv_ld_int(const LVals<FP>& lv, u32 slot) {
    IntVal<FP> r;
    const u64* p = lv.v + (size_t)slot * LVals<FP>::W;   // 16-byte aligned: W is even
#pragma unroll
    for (int i = 0; i < FP::L; i++) {
        u64x2 q = l_ld16(p + 2 * i);
        r.l[i].v[0] = q.x;
        r.l[i].v[1] = q.y;
    }
    u64x2 a = l_ld16(p + 2 * FP::L), b = l_ld16(p + 2 * FP::L + 2);
    r.native.v[0] = a.x; r.native.v[1] = a.y; r.native.v[2] = b.x; r.native.v[3] = b.y;
    return r;
}
template <class FP>
WI_INLINE Fe lv_ld_fe(const LVals<FP>& lv, u32 slot) {
    Fe r;
    {
        const u64* p = lv.v + (size_t)slot * LVals<FP>::W;
        u64x2 a = l_ld16(p), b = l_ld16(p + 2);
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b.x; r.v[3] = b.y;
    }
    return r;
}
// operands that come from cells: written by other kernels, or by an H2E_V_FULL step of this wave (behind a fence)
template <class FP>
WI_INLINE IntVal<FP> l_src_int(const LVals<FP>& lv, const LC& c, const VHdr& h, int which, const u32* lrefs) {
    u32 kind = (h.w[7] >> (3 * which)) & 7u, word = h.w[2 + which];
    if (kind == H2E_VSRC_INT_SLOT) return lv_ld_int<FP>(lv, word);
    u32 refs[FP::L + 1];
#pragma unroll
    for (int j = 0; j <= FP::L; j++) refs[j] = lrefs[word + j];
    return ld_int<FP>(c, refs);
}
template <class FP>
WI_INLINE Fe l_src_fe(const LVals<FP>& lv, const LC& c, const VHdr& h, int which) {
    u32 kind = (h.w[7] >> (3 * which)) & 7u, word = h.w[2 + which];
    if (kind == H2E_VSRC_FE_SLOT) return lv_ld_fe<FP>(lv, word);
    return ld_fe(c, word);
}
template <class FP>
WI_INLINE void l_out_int(const LVals<FP>& lv, const LC& c, const VHdr& h, bool mul_like, const Limb* l, const Fe& native) {
    if ((h.w[0] >> 8) & H2E_VFLAG_STORE) {
        if (mul_like) {
#pragma unroll
            for (int i = 0; i < FP::L; i++) stR(c, h.w[6] + 3 * i, 0, fe_of(l[i]));
            stB(c, h.w[5], 4, native);
        } else {
#pragma unroll
            for (int i = 0; i < FP::L; i++) stB(c, h.w[5] + i, 4, fe_of(l[i]));
            stB(c, h.w[5] + FP::L, 4, native);
        }
    }
    u32 dst = h.w[0] >> 16;
    if (dst != 0xffffu) {
        u64* p = lv.v + (size_t)dst * LVals<FP>::W;
#pragma unroll
        for (int i = 0; i < FP::L; i++) l_st16(p + 2 * i, l[i].v[0], l[i].v[1]);
        l_st16(p + 2 * FP::L, native.v[0], native.v[1]);
        l_st16(p + 2 * FP::L + 2, native.v[2], native.v[3]);
    }
}
template <class FP>
WI_INLINE void l_out_w(const LVals<FP>& lv, const LC& c, const VHdr& h, const Wd<FP::WW>& x) {
    Limb l[FP::L];
    split_limbs<FP>(x, l);
    l_out_int<FP>(lv, c, h, true, l, native_of_w<FP>(c, x));
}
template <class FP>
WI_INLINE void l_out_fe(const LVals<FP>& lv, const LC& c, const VHdr& h, const Fe& v) {
    if ((h.w[0] >> 8) & H2E_VFLAG_STORE) stB(c, h.w[5], 4, v);
    u32 dst = h.w[0] >> 16;
    if (dst != 0xffffu) {
        u64* p = lv.v + (size_t)dst * LVals<FP>::W;
        l_st16(p, v.v[0], v.v[1]);
        l_st16(p + 2, v.v[2], v.v[3]);
    }
}
// the light ops of a step (additions, selections, conditions); `opc` may differ from lane to lane (H2E_VFLAG_MIXED steps)
template <class FP>
WI_INLINE void exec_lop_light(const LVals<FP>& lv, const LC& c, u32 opc, const VHdr& h, const u32* lrefs) {
    constexpr int L = FP::L;
    u32 imm = h.w[1];
    if (opc == H2E_V_SUB) {
        IntVal<FP> a = l_src_int<FP>(lv, c, h, 0, lrefs), b = l_src_int<FP>(lv, c, h, 1, lrefs);
        Limb s[L];
#pragma unroll
        for (int i = 0; i < L; i++) s[i] = wd_sub<2>(wd_add<2>(a.l[i], ceil_limb(c, imm, i)), b.l[i]);
        l_out_int<FP>(lv, c, h, false, s, addmod_n(c, submod_n(c, a.native, b.native), ceil_native(c, imm)));
    } else if (opc == H2E_V_ADD) {
        IntVal<FP> a = l_src_int<FP>(lv, c, h, 0, lrefs), b = l_src_int<FP>(lv, c, h, 1, lrefs);
        Limb s[L];
#pragma unroll
        for (int i = 0; i < L; i++) s[i] = wd_add<2>(a.l[i], b.l[i]);
        l_out_int<FP>(lv, c, h, false, s, addmod_n(c, a.native, b.native));
    } else if (opc == H2E_V_NEG) {
        IntVal<FP> a = l_src_int<FP>(lv, c, h, 0, lrefs);
        Limb s[L];
#pragma unroll
        for (int i = 0; i < L; i++) s[i] = wd_sub<2>(ceil_limb(c, imm, i), a.l[i]);
        l_out_int<FP>(lv, c, h, false, s, submod_n(c, ceil_native(c, imm), a.native));
    } else if (opc == H2E_V_MASK) {
        IntVal<FP> a = l_src_int<FP>(lv, c, h, 0, lrefs);
        Fe coeff = l_src_fe<FP>(lv, c, h, 1);
        bool keep = !wd_is_zero<4>(coeff);
        Limb s[L];
#pragma unroll
        for (int i = 0; i < L; i++) s[i] = keep ? a.l[i] : wd_zero<2>();
        l_out_int<FP>(lv, c, h, false, s, keep ? a.native : wd_zero<4>());
    } else if (opc == H2E_V_BISEC_INT) {
        Fe cond = l_src_fe<FP>(lv, c, h, 0);
        IntVal<FP> a = l_src_int<FP>(lv, c, h, 1, lrefs), b = l_src_int<FP>(lv, c, h, 2, lrefs);
        bool take_a = !wd_is_zero<4>(cond);
        Limb s[L];
#pragma unroll
        for (int i = 0; i < L; i++) s[i] = take_a ? a.l[i] : b.l[i];
        l_out_int<FP>(lv, c, h, false, s, take_a ? a.native : b.native);
    } else if (opc == H2E_V_IS_ZERO) {
        IntVal<FP> a = l_src_int<FP>(lv, c, h, 0, lrefs);
        bool all_zero = true;
#pragma unroll
        for (int i = 0; i < L; i++) all_zero = all_zero && wd_is_zero<2>(a.l[i]);
        bool is_w = wd_eq<4>(a.native, wd_load<4>(c.fc->w_native));
#pragma unroll
        for (int i = 0; i < FP::PW; i++) is_w = is_w && wd_eq<2>(a.l[i], wd_load<2>(c.fc->w_limbs[i]));
        l_out_fe<FP>(lv, c, h, fe_u64((all_zero || is_w) ? 1 : 0));
    } else if (opc == H2E_V_NOT) {
        l_out_fe<FP>(lv, c, h, submod_n(c, fe_u64(1), l_src_fe<FP>(lv, c, h, 0)));
    } else if (opc == H2E_V_AND || opc == H2E_V_OR || opc == H2E_V_XNOR) {
        Fe a = l_src_fe<FP>(lv, c, h, 0), b = l_src_fe<FP>(lv, c, h, 1);
        u64 r = opc == H2E_V_AND ? (a.v[0] & b.v[0]) : opc == H2E_V_OR ? (a.v[0] | b.v[0]) : (1 ^ a.v[0] ^ b.v[0]);
        l_out_fe<FP>(lv, c, h, fe_u64(r));
    }
}
// one lane's op of a step; `opc` is the step's opcode (wave-uniform), everything else is per lane
template <class FP>
WI_INLINE void exec_lop(const LVals<FP>& lv, const LC& c, u32 opc, const VHdr& h, const u32* lrefs) {
    constexpr int L = FP::L;
    u32 imm = h.w[1];
    if (opc == H2E_V_MUL) {
        IntVal<FP> a = l_src_int<FP>(lv, c, h, 0, lrefs), b = l_src_int<FP>(lv, c, h, 1, lrefs);
        Wd<FPX<FP>::AW> A = compose<FP, FPX<FP>::AW>(a.l), B = compose<FP, FPX<FP>::AW>(b.l);
        Wd<FPX<FP>::QW> dq;
        Wd<FP::WW> rem;
        divrem_w<FP>(c, wd_resize<FPX<FP>::XW>(wd_mul<FPX<FP>::AW, FPX<FP>::AW, FPX<FP>::AL, FPX<FP>::AL>(A, B)), dq, rem);
        l_out_w<FP>(lv, c, h, rem);
    } else if (opc == H2E_V_REDUCE) {
        IntVal<FP> a = l_src_int<FP>(lv, c, h, 0, lrefs);
        Wd<FP::WW> rem;
        u64 dsmall;
        divrem_small<FP>(c, compose<FP, FPX<FP>::AW>(a.l), dsmall, rem);
        l_out_w<FP>(lv, c, h, rem);
    } else if (opc == H2E_V_MUL_SMALL) {
        IntVal<FP> a = l_src_int<FP>(lv, c, h, 0, lrefs);
        Wd<1> k = wd_from_u64<1>(imm);
        Limb s[L];
#pragma unroll
        for (int i = 0; i < L; i++) s[i] = wd_resize<2>(wd_mul<2, 1>(a.l[i], k));
        l_out_int<FP>(lv, c, h, false, s, mod_n<5>(c, wd_mul<4, 1>(a.native, k)));
    } else if (opc == H2E_V_DIV) {
        IntVal<FP> b = l_src_int<FP>(lv, c, h, 0, lrefs), a = l_src_int<FP>(lv, c, h, 1, lrefs);
        Wd<FPX<FP>::QW> q0;
        Wd<FP::WW> a_red, b_red, cv;
        divrem_w<FP>(c, wd_resize<FPX<FP>::XW>(compose<FP, FPX<FP>::AW>(a.l)), q0, a_red);
        divrem_w<FP>(c, wd_resize<FPX<FP>::XW>(compose<FP, FPX<FP>::AW>(b.l)), q0, b_red);
        Wd<FP::WW> binv = wd_inv_mod<FP::WW>(b_red, wd_load<FP::WW>(c.fc->w));
        divrem_w<FP>(c, wd_resize<FPX<FP>::XW>(wd_mul<FP::WW, FP::WW>(a_red, binv)), q0, cv);
        l_out_w<FP>(lv, c, h, cv);
    } else if (opc == H2E_V_HINT) {   // the canonical result comes from the predictors (one load per lane)
        u32 slot = imm + ((((h.w[0] >> 8) & H2E_VFLAG_HINT_STRIDED) != 0) ? c.strand * c.hint_stride : 0);
        l_out_w<FP>(lv, c, h, ws_load<FP::WW>(c.hints + (size_t)slot * c.ws));
    } else if (opc == H2E_V_CONST) {
        Wd<FP::WW> x = g_load<FP::WW>(c.pool + imm);
        Limb l[L];
        split_limbs<FP>(x, l);
        Fe native = mod_n<FP::WW>(c, x);
        if ((h.w[0] >> 8) & H2E_VFLAG_STORE) {
#pragma unroll
            for (int i = 0; i < L; i++) stB(c, h.w[5] + i, 0, fe_of(l[i]));
            stB(c, h.w[5] + L, 0, native);
        }
        VHdr h2 = h;
        h2.w[0] &= ~(H2E_VFLAG_STORE << 8);
        l_out_int<FP>(lv, c, h2, false, l, native);
    } else {
        exec_lop_light<FP>(lv, c, opc, h, lrefs);
    }
}

template <class FP>
__global__ void __launch_bounds__(64 * H2E_LEVEL_WAVES) h2e_replay_levels(H2ELaunch L, const InstanceDesc* inst, u32 n_instances) {
    // l_pair: the workgroup replays two (instance, strand) units - lanes 0-31 of every wave the first, lanes 32-63 the
    // second, each with its own value slots; a unit past the end repeats the last one (same values to the same cells)
    u32 lane = threadIdx.x & 63u;
    u32 half = L.l_pair ? (lane >> 5) : 0u;
    u32 unit = min(blockIdx.x * (L.l_pair ? 2u : 1u) + half, n_instances * L.n_strands - 1u);
    u32 instance = unit / L.n_strands, strand = unit % L.n_strands;
    const u32 lead_mask = L.l_pair ? 31u : 63u;   // the lane that runs a tape op (V_FULL) for its unit
    InstanceDesc d = inst[instance];
    LC c;
    c.base = d.base;
    c.range = d.range;
    c.select = d.select;
    c.inputs = d.inputs;
    c.status = d.status;
    c.ob = L.strand_base0 + strand * L.delta_base;
    c.orr = L.strand_range0 + strand * L.delta_range;
    c.os = L.strand_select0 + strand * L.delta_select;
    c.params = L.params + (size_t)strand * L.n_params;
    c.aux = L.aux;
    c.pool = L.const_pool;
    c.fc = &g_fc[FP::ID];
    c.strand = strand;
    c.input_stride = L.input_stride;
    c.sw = L.slot_words;
    c.hints = d.hints;
    c.ws = d.ws;
    c.hint_stride = L.hint_stride;
    c.sel = d.sel;
    c.sel_stride = L.sel_stride;
    c.hs = inst[0].hs;
    extern __shared__ ulonglong2 l_dyn[];
    LVals<FP> lv;
    lv.v = (u64*)l_dyn + (size_t)half * L.l_slots * LVals<FP>::W;
    c.active = true;
    __builtin_amdgcn_s_setprio(3);
    // rounds: every wave runs one step (64 records, one opcode) of the current level, then a barrier; the waves share
    // the instance's value slots
    auto fetch = [&](u32 round, uint4* r) {
        const uint4* p = (const uint4*)(L.lrecs + (size_t)round * (64 * H2E_LEVEL_WAVES) + threadIdx.x);
        r[0] = p[0];
        r[1] = p[1];
    };
    uint4 nxt[2];
    fetch(0, nxt);
    for (u32 round = 0; round < L.l_steps; round++) {
        VHdr h;
        h.w[0] = nxt[0].x; h.w[1] = nxt[0].y; h.w[2] = nxt[0].z; h.w[3] = nxt[0].w;
        h.w[4] = nxt[1].x; h.w[5] = nxt[1].y; h.w[6] = nxt[1].z; h.w[7] = nxt[1].w;
        if (round + 1 < L.l_steps) fetch(round + 1, nxt);   // the next round's records are in flight while this one runs
        u32 w0 = __builtin_amdgcn_readfirstlane(h.w[0]);     // lane 0 of the wave's step
        u32 opc = w0 & 0xffu;
        if (opc == H2E_V_FULL || ((w0 >> 8) & H2E_VFLAG_FENCE)) {
            // a tape op that goes through cells, run by wave 0 for the instance: its rows are its results.  What it reads
            // may have been stored by any wave in earlier rounds, and later rounds read its rows: fences on both sides.
            __threadfence();
            __syncthreads();
            if (opc == H2E_V_FULL) {
                H2EOp op = L.tape[__builtin_amdgcn_readfirstlane(h.w[1])];
                op.opcode = (uint16_t)__builtin_amdgcn_readfirstlane(op.opcode);
                c.active = (lane & lead_mask) == 0;
                exec_op<FP, false>(c, op);
                c.active = true;
                __threadfence();
            }
        } else if ((w0 >> 8) & H2E_VFLAG_MIXED) {
            exec_lop_light<FP>(lv, c, h.w[0] & 0xffu, h, L.lrefs);   // light ops of any mix: every lane its own opcode
        } else if ((h.w[0] & 0xffu) != H2E_V_NOP) {
            exec_lop<FP>(lv, c, opc, h, L.lrefs);   // (a round without its op costs 0.44 us; with it 3.0 us on average)
        }
        lds_round_barrier_workgroup();   // the round's values are in their slots
    }
}

// Wave-mode level replay: ONE wave replays one (instance, strand); a round is up to 64 independent ops of one cost class
// (host: schedule_classes in h2e_capi.cpp), values live in the wave's LDS slots.  What the four-wave kernel above paid
// per round - a workgroup barrier, and a wait for every global store of the round before the next round's records (and
// the per-lane ceil-table constants) could be read: loads and stores share one in-order counter - is gone:
//  * the compact record stream goes through two LDS chunk buffers, filled by LDS-DMA one chunk ahead; the wave waits
//    for memory once per chunk (H2E_WCHUNK records, some 20 rounds), not once per round;
//  * the ceil tables sit in LDS; nothing in a round's body loads from global memory (a few hundred ops of a pairing
//    check read cells of other kernels - they do wait);
//  * result cells that later kernels need are stored and never waited for.
#ifdef H2E_AB_KERNELS   // (A/B kernel: H2E_LEVEL_MODE=wave in -DH2E_DEBUG_HOOKS builds of the C-ABI layer; exp/README.md)
template <class FP>
__global__ void __launch_bounds__(64) h2e_replay_wave(H2ELaunch L, const InstanceDesc* inst, u32 n_instances) {
    const u32 lane = threadIdx.x;
    u32 unit = blockIdx.x;
    u32 instance = unit / L.n_strands, strand = unit % L.n_strands;
    InstanceDesc d = inst[instance];
    LC c;
    c.base = d.base;
    c.range = d.range;
    c.select = d.select;
    c.inputs = d.inputs;
    c.status = d.status;
    c.ob = L.strand_base0 + strand * L.delta_base;
    c.orr = L.strand_range0 + strand * L.delta_range;
    c.os = L.strand_select0 + strand * L.delta_select;
    c.params = L.params + (size_t)strand * L.n_params;
    c.aux = L.aux;
    c.pool = L.const_pool;
    c.fc = &g_fc[FP::ID];
    c.strand = strand;
    c.input_stride = L.input_stride;
    c.sw = L.slot_words;
    c.hints = d.hints;
    c.ws = d.ws;
    c.hint_stride = L.hint_stride;
    c.sel = d.sel;
    c.sel_stride = L.sel_stride;
    c.hs = inst[0].hs;
    c.active = true;
    extern __shared__ ulonglong2 w_dyn[];
    H2EVRec* rbuf = (H2EVRec*)w_dyn;                                   // [2][H2E_WCHUNK]
    u64* ceil_tab = (u64*)(rbuf + 2 * H2E_WCHUNK);                      // [64][H2E_MAX_L][2] + [64][4]
    constexpr u32 CEIL_WORDS = 64 * H2E_MAX_L * 2 + 64 * 4;
    LVals<FP> lv;
    lv.v = ceil_tab + CEIL_WORDS;
    for (u32 i = lane; i < 64 * H2E_MAX_L * 2; i += 64) l_st8(ceil_tab + i, ((const u64*)c.fc->ceil_limbs)[i]);
    for (u32 i = lane; i < 64 * 4; i += 64) l_st8(ceil_tab + 64 * H2E_MAX_L * 2 + i, ((const u64*)c.fc->ceil_native)[i]);
    c.ceil_lds = ceil_tab;
    __builtin_amdgcn_s_setprio(3);
    const u32 n_chunks = L.l_recs / H2E_WCHUNK;
    auto load_chunk = [&](u32 chunk) {   // H2E_WCHUNK records = 8 KB: 8 LDS-DMA pieces of 16 bytes per lane
        const char* src = (const char*)(L.lrecs + (size_t)chunk * H2E_WCHUNK);
        char* dst = (char*)(rbuf + (size_t)(chunk & 1u) * H2E_WCHUNK);
#pragma unroll
        for (u32 k = 0; k < H2E_WCHUNK * 32u / 1024u; k++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k * 1024u + lane * 16u),
                                             (__attribute__((address_space(3))) void*)(dst + k * 1024u), 16, 0, 0);
    };
    load_chunk(0);
    if (n_chunks > 1) load_chunk(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    u32 cur_chunk = 0;
    const uint2* rounds = (const uint2*)L.lrounds;
    for (u32 round = 0; round < L.l_steps; round++) {
        uint2 rd = rounds[round];                      // wave-uniform: scalar loads
        u32 first = __builtin_amdgcn_readfirstlane(rd.x), meta = __builtin_amdgcn_readfirstlane(rd.y);
        u32 cnt = meta & 0xffu, kind = meta >> 8;
        u32 chunk = first / H2E_WCHUNK;
        if (chunk != cur_chunk) {
            // this chunk was requested a chunk ago: the wait is for it and for the stores still under way
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur_chunk = chunk;
            if (chunk + 1 < n_chunks) load_chunk(chunk + 1);   // into the buffer the wave has just left
        }
        if (kind == H2E_V_FULL) {
            // a tape op that goes through cells, run by lane 0: what it reads may have been stored in earlier rounds, and
            // later rounds read its rows
            const H2E_AS_LDS u32* rp = (const H2E_AS_LDS u32*)(rbuf + (size_t)(chunk & 1u) * H2E_WCHUNK + first % H2E_WCHUNK);
            u32 op_index = __builtin_amdgcn_readfirstlane(rp[1]);
            __threadfence();
            H2EOp op = L.tape[op_index];
            op.opcode = (uint16_t)__builtin_amdgcn_readfirstlane(op.opcode);
            c.active = lane == 0;
            exec_op<FP, false>(c, op);
            c.active = true;
            __threadfence();
        } else if (lane < cnt) {
            const H2E_AS_LDS u32x4* rp = (const H2E_AS_LDS u32x4*)(rbuf + (size_t)(chunk & 1u) * H2E_WCHUNK + first % H2E_WCHUNK + lane);
            u32x4 a = rp[0], b = rp[1];
            VHdr h;
            h.w[0] = a.x; h.w[1] = a.y; h.w[2] = a.z; h.w[3] = a.w;
            h.w[4] = b.x; h.w[5] = b.y; h.w[6] = b.z; h.w[7] = b.w;
            if (kind == 0) exec_lop_light<FP>(lv, c, h.w[0] & 0xffu, h, L.lrefs);
            else exec_lop<FP>(lv, c, kind, h, L.lrefs);
        }
        lds_round_barrier_wave();   // (one wave: only the LDS accesses of the round are ordered; global stores stay in flight)
    }
}
#endif

// ------------------------------------------------------------------------------------------------
// Hint store (field_chain.hpp HintStore): the cells the full expansion needs in place, straight from the hint slots.
// One lane per (store op, instance), instances minor: the lanes of a wave read the same hint slots and write the same
// cells of consecutive instances (contiguous runs, like the expansion).
// Leaves of a store record (field_chain.hpp): kind 0 hint slot, 1 pool word offset, 2 input slot, 3 = entry `index` of the
// launch's extension table (H2ELaunch::s_ext, 8 words each: type, arguments) - what the forked MSM segments need:
//   H2E_SX_HINT   a0 = hint slot of strand 0 (+ strand * hint_stride)
//   H2E_SX_SEL    a0 = selection-buffer entry of strand 0 (+ strand * sel_stride), a1 = 0: x, 1: y (canonical, from the select pre-kernel)
//   H2E_SX_INPUT  a0 = input slot of strand 0 (+ strand * input_stride)
//   H2E_SX_CELLS  an integer that lives in cells written before this launch: a0 .. a(L-1) limb cells, aL native cell
//                 (references like an op's: absolute, strand-relative or through the strand's parameters)
template <class FP>
WI_INLINE Wd<FP::WW> hs_leaf(const LC& c, u32 t, const u32* ext) {
    u32 kind = t >> 30, index = t & 0x3fffffu;
    if (kind == 0) return ws_load<FP::WW>(c.hints + (size_t)index * c.ws);
    if (kind == 1) return g_load<FP::WW>(c.pool + index);
    if (kind == 2) return g_load<FP::WW>(c.inputs + (size_t)index * c.sw);
    const u32* e = ext + (size_t)index * H2E_SX_WORDS;
    u32 type = e[0];
    if (type == H2E_SX_HINT) return ws_load<FP::WW>(c.hints + (size_t)(e[1] + c.strand * c.hint_stride) * c.ws);
    if (type == H2E_SX_SEL) return ws_load<FP::WW>(c.sel + ((size_t)H2E_SEL_SLOTS * (e[1] + c.strand * c.sel_stride) + e[2]) * c.ws);
    return g_load<FP::WW>(c.inputs + (size_t)(e[1] + c.strand * c.input_stride) * c.sw);
}
template <class FP>
__global__ void __launch_bounds__(64, H2E_CHAIN_WAVES) h2e_hint_store(H2ELaunch L, const InstanceDesc* inst, u32 n_instances) {
    constexpr int NL = FP::L;
    // lanes: [store op][strand][instance], instances minor (a forked segment's store ops run once per strand, at the strand's
    // row offsets, with its hint / selection / input slots and parameters)
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    u32 total = L.n_sops * L.n_strands * n_instances;
    bool active = gid < total;
    if (!active) gid = total - 1;
    u32 instance = gid % n_instances, strand = (gid / n_instances) % L.n_strands, sop = gid / n_instances / L.n_strands;
    InstanceDesc d = inst[instance];
    LC c;
    c.base = d.base;
    c.range = d.range;
    c.select = d.select;
    c.inputs = d.inputs;
    c.status = d.status;
    c.ob = L.strand_base0 + strand * L.delta_base;
    c.orr = L.strand_range0 + strand * L.delta_range;
    c.os = L.strand_select0 + strand * L.delta_select;
    c.params = L.params + (size_t)strand * L.n_params;
    c.aux = L.aux;
    c.pool = L.const_pool;
    c.fc = &g_fc[FP::ID];
    c.strand = strand;
    c.input_stride = L.input_stride;
    c.sw = L.slot_words;
    c.hints = d.hints;
    c.ws = d.ws;
    c.hint_stride = L.hint_stride;
    c.sel = d.sel;
    c.sel_stride = L.sel_stride;
    c.hs = inst[0].hs;
    c.active = active;
    const u32* ext = L.s_ext;
    const u32* rec = L.s_words + L.s_offsets[sop];
    u32 w0 = rec[0], row = rec[1], rrow = rec[2];
    u32 kind = w0 & 0xffu, n_terms = (w0 >> 8) & 0xffu, kidx = w0 >> 16;
    if (kind == H2E_S_W) {   // a mul-like result: limbs in its range rows, native in its base row
        Wd<FP::WW> x = hs_leaf<FP>(c, rec[3], ext);
        Limb l[NL];
        split_limbs<FP>(x, l);
        if (active) {
#pragma unroll
            for (int i = 0; i < NL; i++) stR(c, rrow + 3 * i, 0, fe_of(l[i]));
            stB(c, row, 4, native_of_w<FP>(c, x));
        }
    } else if (kind == H2E_S_LIN) {
        // limbs modulo 2^128 (the true values are non-negative and below 2^114), native as a non-negative sum reduced mod n
        Limb acc[NL];
        const u64* kt = L.s_ktab + (size_t)kidx * (2 * NL + 4);
#pragma unroll
        for (int i = 0; i < NL; i++) {
            acc[i].v[0] = g_ld8(kt + 2 * i);
            acc[i].v[1] = g_ld8(kt + 2 * i + 1);
        }
        Wd<5> nat = wd_resize<5>(g_load<4>(kt + 2 * NL));
        Fe n = n_of(c);
        // Terms in batches of TB: the term words, then their extension entries, then the leaves themselves - three rounds of
        // independent loads per batch instead of three dependent loads per term (a windows' record has ~10 terms, and a dependent
        // load costs microseconds next to a running expansion).  A padding term has coefficient 0; every load of a batch is
        // unconditional (a leaf that is not a W value - an integer in cells - or a padding term reads hint slot 0 and ignores it).
        constexpr u32 TB = 6;
        for (u32 t0 = 0; t0 < n_terms; t0 += TB) {
            u32 tw[TB], e0[TB], e1[TB], e2[TB];
#pragma unroll
            for (u32 k = 0; k < TB; k++) tw[k] = t0 + k < n_terms ? rec[3 + t0 + k] : (128u << 22);
#pragma unroll
            for (u32 k = 0; k < TB; k++) {
                const u32* e = ext + (size_t)((tw[k] >> 30) == 3u ? (tw[k] & 0x3fffffu) : 0u) * H2E_SX_WORDS;
                e0[k] = e[0];
                e1[k] = e[1];
                e2[k] = e[2];
            }
            Wd<FP::WW> x[TB];
            bool cells[TB];
#pragma unroll
            for (u32 k = 0; k < TB; k++) {
                u32 kind = tw[k] >> 30, index = tw[k] & 0x3fffffu;
                const u64* p = c.hints + (size_t)index * c.ws;                                   // kind 0: hint slot
                if (kind == 1) p = c.pool + index;
                if (kind == 2) p = c.inputs + (size_t)index * c.sw;
                cells[k] = kind == 3u && e0[k] == H2E_SX_CELLS;
                if (kind == 3u) {
                    p = c.hints + (size_t)(e1[k] + c.strand * c.hint_stride) * c.ws;             // H2E_SX_HINT
                    if (e0[k] == H2E_SX_SEL) p = c.sel + ((size_t)H2E_SEL_SLOTS * (e1[k] + c.strand * c.sel_stride) + e2[k]) * c.ws;
                    if (e0[k] == H2E_SX_INPUT) p = c.inputs + (size_t)(e1[k] + c.strand * c.input_stride) * c.sw;
                    if (cells[k]) p = c.hints;
                }
                x[k] = g_load<FP::WW>(p);
            }
#pragma unroll
            for (u32 k = 0; k < TB; k++) {
                int coef = (int)((tw[k] >> 22) & 0xffu) - 128;
                Limb l[NL];
                Fe xn;
                if (cells[k]) {
                    // an integer from outside the segment, as its cells hold it (its limbs need not be the canonical split)
                    const u32* e = ext + (size_t)(tw[k] & 0x3fffffu) * H2E_SX_WORDS + 1;
#pragma unroll
                    for (int i = 0; i < NL; i++) l[i] = ld_limb(c, e[i]);
                    xn = ld_fe(c, e[NL]);
                } else {
                    split_limbs<FP>(x[k], l);
                    xn = native_of_w<FP>(c, x[k]);
                }
                u32 m = (u32)(coef < 0 ? -coef : coef);
#pragma unroll
                for (int i = 0; i < NL; i++) {
                    Limb p = wd_resize<2>(wd_mul_small<2>(l[i], m));
                    acc[i] = coef < 0 ? wd_sub<2>(acc[i], p) : wd_add<2>(acc[i], p);
                }
                if (coef < 0) xn = wd_sub<4>(n, xn);   // -x = n - x (in (0, n])
                wd_mac_small<4>(nat, xn, m);
            }
        }
        // nat < 2^12 n (the host rejects a combination whose |coefficients| sum to 4096 or more, field_chain.hpp HintStore::compile:
        // the top word taken below is 64 bits): small-quotient reduction
        Fe natr;
        {
            constexpr int SH = NK - 52;
            u64 a_top = wd_shr<1, SH>(nat).v[0];
            double inv = 1.0 / (double)(wd_shr<1, SH>(n).v[0] + 1);
            u64 qe = (u64)((double)a_top * inv);
            qe = qe > 0 ? qe - 1 : 0;
            Wd<5> ne = wd_resize<5>(n);
            Wd<5> r = wd_sub<5>(nat, wd_mul_small<4>(n, (u32)qe));
#pragma unroll
            for (int it = 0; it < 4; it++) {
                bool ge = wd_geq<5>(r, ne);
                r = wd_select<5>(ge, wd_sub<5>(r, ne), r);
            }
            natr = wd_resize<4>(r);
        }
        if (active) {
#pragma unroll
            for (int i = 0; i < NL; i++) stB(c, row + i, 4, fe_of(acc[i]));
            stB(c, row + NL, 4, natr);
        }
    } else if (kind == H2E_S_FE) {
        Wd<FP::WW> x = hs_leaf<FP>(c, rec[3], ext);
        if (active) stB(c, row, 4, fe_u64(x.v[0] & 1));
    } else if (kind == H2E_S_CONST) {   // assign_int_constant: limb i in (row + i, col 0), native in (row + L, col 0)
        Wd<FP::WW> x = hs_leaf<FP>(c, rec[3], ext);
        Limb l[NL];
        split_limbs<FP>(x, l);
        Fe native = mod_n<FP::WW>(c, x);
        if (active) {
#pragma unroll
            for (int i = 0; i < NL; i++) stB(c, row + i, 0, fe_of(l[i]));
            stB(c, row + NL, 0, native);
        }
    } else if (kind == H2E_S_FULL) {
        H2EOp op = L.tape[row];
        exec_op<FP, false>(c, op);
    }
}

// ================================================================================================
// Montgomery arithmetic (R = 2^(64 N)) — used by the value-predictor kernels (mod w) and the inverse fix-up (mod n)
template <int N>
struct Mont {
    Wd<N> p, r2, r1;
    u64 minv;
};
WI_INLINE void mac64(u64 a, u64 b, u64 c, u64& carry, u64& out) {  // out = low(a*b + c + carry); carry = high
    u64 lo, hi;
    mul_wide64(a, b, lo, hi);
    lo += c;
    hi += (lo < c);
    lo += carry;
    hi += (lo < carry);
    out = lo;
    carry = hi;
}
// Product scanning (FIPS) over 32-bit limbs with a 96-bit column accumulator: every partial product is one
// v_mad_u64_u32 into the low 64 bits plus one add-with-carry into the third word, and nothing else - the operand
// scanning (CIOS) form the compiler was given before spent as many instructions again on zero-extending and
// re-pairing its 32-bit carries (600 instructions per multiplication against 330; 78 -> 108 G multiplications/s on
// the whole device, exp/mm_bench).  a, b < p, p odd and below 2^(64 N) (the sum a b + m p may need bit 64 N: the third
// accumulator word carries it into the final comparison); result < p.
template <int N>
WI_INLINE Wd<N> mont_mul(const Mont<N>& M, const Wd<N>& a, const Wd<N>& b) {
    constexpr int L32 = 2 * N;
    u32 m[L32], r[L32];
    u32 minv = (u32)M.minv;
    WdAcc A{0, 0};
#pragma unroll
    for (int k = 0; k < L32; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) wd_mac(A, limb32<N>(a, i), limb32<N>(b, k - i));
#pragma unroll
        for (int i = 0; i < k; i++) wd_mac(A, m[i], limb32<N>(M.p, k - i));
        m[k] = (u32)A.lo * minv;
        wd_mac(A, m[k], limb32<N>(M.p, 0));
        (void)wd_acc_shift(A);
    }
#pragma unroll
    for (int k = L32; k < 2 * L32; k++) {
#pragma unroll
        for (int i = k - L32 + 1; i < L32; i++) wd_mac(A, limb32<N>(a, i), limb32<N>(b, k - i));
#pragma unroll
        for (int i = k - L32 + 1; i < L32; i++) wd_mac(A, m[i], limb32<N>(M.p, k - i));
        r[k - L32] = wd_acc_shift(A);
    }
    Wd<N> o;
#pragma unroll
    for (int i = 0; i < N; i++) o.v[i] = (u64)r[2 * i] | ((u64)r[2 * i + 1] << 32);
    bool ge = (u32)A.lo != 0 || wd_geq<N>(o, M.p);
    return ge ? wd_sub<N>(o, M.p) : o;
}
template <int N>
WI_INLINE Wd<N> mont_add(const Mont<N>& M, const Wd<N>& a, const Wd<N>& b) {
    u64 c;
    Wd<N> s = wd_add_c<N>(a, b, c);
    return (c || wd_geq<N>(s, M.p)) ? wd_sub<N>(s, M.p) : s;
}
template <int N>
WI_INLINE Wd<N> mont_sub(const Mont<N>& M, const Wd<N>& a, const Wd<N>& b) {
    return wd_geq<N>(a, b) ? wd_sub<N>(a, b) : wd_sub<N>(wd_add<N>(a, M.p), b);
}
template <int N>
WI_INLINE Wd<N> mont_dbl(const Mont<N>& M, const Wd<N>& a) { return mont_add<N>(M, a, a); }
template <int N>
WI_INLINE Wd<N> to_mont(const Mont<N>& M, const Wd<N>& a) { return mont_mul<N>(M, a, M.r2); }
template <int N>
WI_INLINE Wd<N> from_mont(const Mont<N>& M, const Wd<N>& a) { return mont_mul<N>(M, a, wd_from_u64<N>(1)); }
// inverse in the Montgomery domain (0 -> 0)
template <int N>
WI_INLINE Wd<N> mont_inv(const Mont<N>& M, const Wd<N>& a) {
    return to_mont<N>(M, wd_inv_mod<N>(from_mont<N>(M, a), M.p));
}
template <class FP>
WI_INLINE Mont<FP::WW> mont_w(const H2EFieldConsts* fc) {
    Mont<FP::WW> M;
    M.p = wd_load<FP::WW>(fc->w);
    M.r2 = wd_load<FP::WW>(fc->w_r2);
    M.r1 = wd_load<FP::WW>(fc->w_r1);
    M.minv = fc->w_minv;
    return M;
}
WI_INLINE Mont<4> mont_n(const H2EFieldConsts* fc) {
    Mont<4> M;
    M.p = wd_load<4>(fc->n);
    M.r2 = wd_load<4>(fc->n_r2);
    M.r1 = wd_load<4>(fc->n_r1);
    M.minv = fc->n_minv;
    return M;
}

// ------------------------------------------------------------------------------------------------
// Fix-up of the is_zero inverse witnesses (base_chip.rs:298-321: b = a^-1 or 0), batched with Montgomery's
// trick: one lane owns FIXUP_K consecutive cells of one strand's list; the destination cells themselves hold
// the running prefix products between the forward and the backward pass.
#ifndef H2E_FIXUP_K
#define H2E_FIXUP_K 64
#endif
static constexpr int FIXUP_K = H2E_FIXUP_K;
__global__ void __launch_bounds__(64, H2E_CHAIN_WAVES) h2e_fixup_inverses(H2ELaunch L, const InstanceDesc* inst, u32 n_instances,
                                                         const H2EFieldConsts* fc) {
    u32 chunks = (L.n_fixups + FIXUP_K - 1) / FIXUP_K;
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    u32 total = n_instances * L.n_strands * chunks;
    if (gid >= total) return;
    // instance minor: the lanes of a wave read / write the same cell of consecutive instances (contiguous)
    u32 instance = gid % n_instances, strand = (gid / n_instances) % L.n_strands, chunk = gid / (n_instances * L.n_strands);
    u64* base = inst[instance].base;
    u32 hs = inst[0].hs;
    u32 ob = L.strand_base0 + strand * L.delta_base;
    Mont<4> M = mont_n(fc);
    u32 lo = chunk * FIXUP_K, hi = min(lo + FIXUP_K, L.n_fixups);
    // both passes fetch FB rows at a time, independent loads in flight together, before the serial multiplications
    // over them.
    constexpr u32 FB = 8;
    // No Montgomery conversions: with mm(a, b) = a b / R on the canonical values themselves, the prefixes are
    // p_i = p_(i-1) x_i / R (p_0 = 1); from u_k = 1 / p_k (a plain inverse) the backward pass gets
    // 1 / x_i = mm(p_(i-1), u_i) and u_(i-1) = mm(x_i, u_i): three multiplications per cell.
    Fe acc = wd_from_u64<4>(1);
    for (u32 i0 = lo; i0 < hi; i0 += FB) {
        u64* rows[FB];
        Fe xs[FB];
#pragma unroll
        for (u32 j = 0; j < FB; j++) rows[j] = base + (size_t)(L.fixups[min(i0 + j, hi - 1)] + ob) * 10 * hs;
#pragma unroll
        for (u32 j = 0; j < FB; j++) xs[j] = ld_cell(rows[j], hs);
#pragma unroll
        for (u32 j = 0; j < FB; j++) {
            if (i0 + j < hi) {
                st_cell(rows[j] + (size_t)2 * hs, hs, acc);  // prefix product of the non-zero values before i
                if (!wd_is_zero<4>(xs[j])) acc = mont_mul<4>(M, acc, xs[j]);
            }
        }
    }
    Fe ainv = wd_inv_mod<4>(acc, M.p);
    u64 dgv[4] = {0, 0, 0, 0};
    for (u32 i1 = hi; i1 > lo;) {
        u32 cnt = min(FB, i1 - lo);   // rows i1-1 ... i1-cnt, in that order
        u64* rows[FB];
        Fe xs[FB], pre[FB];
#pragma unroll
        for (u32 j = 0; j < FB; j++) rows[j] = base + (size_t)(L.fixups[i1 - 1 - min(j, cnt - 1)] + ob) * 10 * hs;
#pragma unroll
        for (u32 j = 0; j < FB; j++) {
            xs[j] = ld_cell(rows[j], hs);
            pre[j] = ld_cell(rows[j] + (size_t)2 * hs, hs);
        }
#pragma unroll
        for (u32 j = 0; j < FB; j++) {
            if (j < cnt) {
                Fe out = wd_zero<4>();
                if (!wd_is_zero<4>(xs[j])) {
                    out = mont_mul<4>(M, ainv, pre[j]);
                    ainv = mont_mul<4>(M, ainv, xs[j]);
                }
                st_cell(rows[j] + (size_t)2 * hs, hs, out);
                if (L.col[0] != nullptr) {   // column emission (h2e_run_columns): the witness also into the instance's base column 1
                    u64* q = L.col[0] + (size_t)instance * L.col_stride[0] + ((size_t)L.col_rows[0] + L.fixups[i1 - 1 - j] + ob) * 4;
                    g_st16(q, out.v[0], out.v[1]);
                    g_st16(q + 2, out.v[2], out.v[3]);
                }
                if (L.dg_out != nullptr) {   // the inverse witness is a cell of the base array: (row, column 1)
                    u32 k0, k1;
                    dg_keys((L.fixups[i1 - 1 - j] + ob) * 5u + 1u, k0, k1);
#pragma unroll
                    for (int w = 0; w < 4; w++) dgv[w] += (u64)(u32)out.v[w] * k0 + (u64)(u32)(out.v[w] >> 32) * k1;
                }
            }
        }
        i1 -= cnt;
    }
    if (L.dg_out != nullptr)
#pragma unroll
        for (int w = 0; w < 4; w++)
            if (dgv[w]) atomicAdd((unsigned long long*)(L.dg_out + ((size_t)(blockIdx.x & (L.dg_shards - 1u)) * 3 * n_instances + instance) * 4 + w), (unsigned long long)dgv[w]);
}

// ------------------------------------------------------------------------------------------------
// Value-predictor ("V") kernels: native Jacobian arithmetic over W (a = 0 curves) in Montgomery form.
template <int N>
struct Jac {
    Wd<N> x, y, z;
};
// The predictors' chains are single waves that run next to the bandwidth-bound expansion of an earlier segment.
// With every field multiplication inlined one loop iteration is ~260 KB of straight-line code that streams
// through the instruction cache from L2 - fine alone, 3.5x slower once L2 is saturated by the other kernel.
// A called multiplication keeps the loop inside the instruction cache.
template <class FP>
struct MontW : Mont<FP::WW> {};
template <class FP>
__device__ __attribute__((noinline)) Wd<FP::WW> mont_mul_w(Wd<FP::WW> a, Wd<FP::WW> b) {
    Mont<FP::WW> M;
    M.p = wd_load<FP::WW>(g_fc[FP::ID].w);
    M.minv = g_fc[FP::ID].w_minv;
    return mont_mul<FP::WW>(M, a, b);
}
template <int N>
WI_INLINE Wd<N> mm(const Mont<N>& M, const Wd<N>& a, const Wd<N>& b) { return mont_mul<N>(M, a, b); }
template <class FP>
WI_INLINE Wd<FP::WW> mm(const MontW<FP>& M, const Wd<FP::WW>& a, const Wd<FP::WW>& b) { return mont_mul_w<FP>(a, b); }
template <class MT, int N>
WI_INLINE Jac<N> jac_dbl(const MT& M, const Jac<N>& p, Wd<N>& num) {  // num = 3 X^2 ; denominator = result z = 2 Y Z
    Wd<N> a = mm(M, p.x, p.x), b = mm(M, p.y, p.y), cc = mm(M, b, b);
    Wd<N> xb = mont_add<N>(M, p.x, b);
    Wd<N> d = mont_dbl<N>(M, mont_sub<N>(M, mont_sub<N>(M, mm(M, xb, xb), a), cc));
    Wd<N> e = mont_add<N>(M, mont_dbl<N>(M, a), a);
    Wd<N> f = mm(M, e, e);
    Jac<N> r;
    r.x = mont_sub<N>(M, f, mont_dbl<N>(M, d));
    Wd<N> c8 = mont_dbl<N>(M, mont_dbl<N>(M, mont_dbl<N>(M, cc)));
    r.y = mont_sub<N>(M, mm(M, e, mont_sub<N>(M, d, r.x)), c8);
    r.z = mont_dbl<N>(M, mm(M, p.y, p.z));
    num = e;
    return r;
}
// p (Jacobian) + q (affine).  num = S2 - Y1 ; denominator = result z = Z1 * H   (lambda = num / z3 in either order)
template <class MT, int N>
WI_INLINE Jac<N> jac_madd(const MT& M, const Jac<N>& p, const Wd<N>& qx, const Wd<N>& qy, Wd<N>& num) {
    Wd<N> z1z1 = mm(M, p.z, p.z);
    Wd<N> u2 = mm(M, qx, z1z1);
    Wd<N> s2 = mm(M, mm(M, qy, p.z), z1z1);
    Wd<N> h = mont_sub<N>(M, u2, p.x), r = mont_sub<N>(M, s2, p.y);
    Wd<N> hh = mm(M, h, h), hhh = mm(M, hh, h), v = mm(M, p.x, hh);
    Jac<N> o;
    o.x = mont_sub<N>(M, mont_sub<N>(M, mm(M, r, r), hhh), mont_dbl<N>(M, v));
    o.y = mont_sub<N>(M, mm(M, r, mont_sub<N>(M, v, o.x)), mm(M, p.y, hhh));
    o.z = mm(M, p.z, h);
    num = r;
    return o;
}
// p + q, both Jacobian.  num = S2 - S1 ; denominator = result z = Z1 Z2 H
template <class MT, int N>
WI_INLINE Jac<N> jac_add(const MT& M, const Jac<N>& p, const Jac<N>& q, Wd<N>& num) {
    Wd<N> z1z1 = mm(M, p.z, p.z), z2z2 = mm(M, q.z, q.z);
    Wd<N> u1 = mm(M, p.x, z2z2), u2 = mm(M, q.x, z1z1);
    Wd<N> s1 = mm(M, mm(M, p.y, q.z), z2z2), s2 = mm(M, mm(M, q.y, p.z), z1z1);
    Wd<N> h = mont_sub<N>(M, u2, u1), r = mont_sub<N>(M, s2, s1);
    Wd<N> hh = mm(M, h, h), hhh = mm(M, hh, h), v = mm(M, u1, hh);
    Jac<N> o;
    o.x = mont_sub<N>(M, mont_sub<N>(M, mm(M, r, r), hhh), mont_dbl<N>(M, v));
    o.y = mont_sub<N>(M, mm(M, r, mont_sub<N>(M, v, o.x)), mm(M, s1, hhh));
    o.z = mm(M, mm(M, p.z, q.z), h);
    num = r;
    return o;
}

struct VC {  // V-kernel lane context
    LC c;
    u64* nd;
    u64* jac;
};
// canonical W value of an assigned (reduced, times == 1) integer from its limb cells -> Montgomery form
template <class FP>
WI_INLINE Wd<FP::WW> ld_w_mont(const LC& c, const Mont<FP::WW>& M, const u32* limb_refs) {
    Limb l[FP::L];
#pragma unroll
    for (int i = 0; i < FP::L; i++) l[i] = ld_limb(c, limb_refs[i]);
    return to_mont<FP::WW>(M, wd_resize<FP::WW>(compose<FP, FPX<FP>::AW>(l)));
}
template <class FP>
WI_INLINE void st_nd(const VC& v, u32 slot, const Wd<FP::WW>& num, const Wd<FP::WW>& den) {
    u64* p = v.nd + (size_t)slot * 2 * v.c.ws;
    ws_store<FP::WW>(p, num);
    ws_store<FP::WW>(p + v.c.ws, den);
}
// record of one ecc op of a fully hinted chain, kept in the nd area of the op's hint block: (numerator, z of the
// result = denominator, Jacobian x, y of the result), Montgomery form
template <class FP>
WI_INLINE void st_rec(const VC& v, u32 block_slot, const Wd<FP::WW>& num, const Wd<FP::WW>& z, const Wd<FP::WW>& x,
                      const Wd<FP::WW>& y) {
    st_nd<FP>(v, block_slot, num, z);
    st_nd<FP>(v, block_slot + 1, x, y);
}
template <class FP>
WI_INLINE void st_jac(const VC& v, u32 slot, const Jac<FP::WW>& p) {
    u64* q = v.jac + (size_t)slot * 3 * v.c.ws;
    ws_store<FP::WW>(q, p.x);
    ws_store<FP::WW>(q + v.c.ws, p.y);
    ws_store<FP::WW>(q + 2 * (size_t)v.c.ws, p.z);
}
template <class FP>
WI_INLINE Jac<FP::WW> ld_jac(const VC& v, u32 slot) {
    const u64* q = v.jac + (size_t)slot * 3 * v.c.ws;
    Jac<FP::WW> p;
    p.x = ws_load<FP::WW>(q);
    p.y = ws_load<FP::WW>(q + v.c.ws);
    p.z = ws_load<FP::WW>(q + 2 * (size_t)v.c.ws);
    return p;
}

// args of the V kernels (uint32 arrays, absolute cell refs unless noted)
//  CANDIDATES: [0] group size sz; then, per lane, a ref table of sz*2*(L+1) point refs + 2*(L+1) init refs taken
//              from the params table (same layout as the X strand's parameters).  hints: 2^sz - 1 per lane.
//              jac scratch: 2^sz slots per lane.
//  WINDOWS:    [0] n_groups, [1] group_size, [2] n_points, [3..3+2(L+1)) refs of -r1 (x limbs, x native, y limbs,
//              y native), then n_groups aux offsets of the candidate tables.  params: bit cell of point j = param j.
//              hints: n_groups per lane.  jac scratch: 1 slot per lane (the window's final sum).
//  TAIL:       [0] windows, [1] odd-groups flag, [2..) refs of r1, then refs of -r2 (2(L+1) each),
//              [last] jac scratch slot of window 0's sum.  hints: windows * (2 + odd).
template <class FP>
__global__ void __launch_bounds__(64, H2E_CHAIN_WAVES) h2e_predict(H2EPreKernel K, const u32* args, const u32* params_all, const u32* aux,
                                                  const InstanceDesc* inst, u32 n_instances, const H2EFieldConsts* fc) {
    constexpr int L = FP::L, NW = FP::WW, NR = 2 * (L + 1);
    __builtin_amdgcn_s_setprio(3);  // value chain: critical path (see h2e_run_tape<.., true>)
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_instances * K.n_lanes) return;
    u32 instance = gid % n_instances, lane = gid / n_instances;   // instance-minor: the workspace slots of a wave are contiguous
    InstanceDesc d = inst[instance];
    VC v;
    v.c.base = d.base;
    v.c.range = d.range;
    v.c.select = d.select;
    v.c.inputs = d.inputs;
    v.c.status = d.status;
    v.c.ob = v.c.orr = v.c.os = 0;
    v.c.hs = inst[0].hs;
    v.c.params = params_all + K.params_begin + (size_t)lane * K.n_params;
    v.c.aux = aux;
    v.c.pool = nullptr;
    v.c.fc = fc;
    v.c.strand = lane;
    v.c.input_stride = 0;
    v.c.hints = nullptr;
    v.c.ws = d.ws;
    v.c.hint_stride = 0;
    v.nd = d.nd;
    v.jac = d.jac;
    const u32* a = args + K.args_begin;
    MontW<FP> M;
    (Mont<NW>&)M = mont_w<FP>(fc);
    u32 hint0 = K.hint_base + lane * K.hints_per_lane;
    if (K.kind == H2E_PRE_MSM_CANDIDATES) {
        // cl[i] = cl[i - lowbit(i)] + pts[ctz(i)]   (ecc_chip.rs:266-272), a = cl[other] Jacobian, b = pts affine
        u32 sz = a[0];
        u32 j0 = K.scratch_begin + lane * (1u << sz);
        const u32* prm = v.c.params;
        // (statically indexed: a loop with a run-time trip count put this table into scratch memory - 336 / 544 bytes per lane - in the
        // run's serial head; a group has at most five points, the ones beyond sz are never selected)
        Wd<NW> px[5], py[5];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const u32 jj = (u32)j < sz ? (u32)j : 0u;
            px[j] = ld_w_mont<FP>(v.c, M, prm + jj * NR);
            py[j] = ld_w_mont<FP>(v.c, M, prm + jj * NR + L + 1);
        }
        Jac<NW> init;
        init.x = ld_w_mont<FP>(v.c, M, prm + sz * NR);
        init.y = ld_w_mont<FP>(v.c, M, prm + sz * NR + L + 1);
        init.z = M.r1;
        st_jac<FP>(v, j0, init);
        for (u32 i = 1; i < (1u << sz); i++) {
            u32 pos = __builtin_ctz(i), other = i - (1u << pos);
            Jac<NW> p = ld_jac<FP>(v, j0 + other);
            Wd<NW> qx = px[0], qy = py[0];
#pragma unroll
            for (int j = 1; j < 5; j++)
                if ((u32)j == pos) {
                    qx = px[j];
                    qy = py[j];
                }
            Wd<NW> num;
            Jac<NW> r = jac_madd(M, p, qx, qy, num);
            st_nd<FP>(v, hint0 + i - 1, num, r.z);
            st_jac<FP>(v, j0 + i, r);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Field-domain predictor chain (field_chain.hpp; tape.h H2E_PRE_FIELD_CHAIN): the canonical value of every mul-like result
// of a pairing check, computed as plain arithmetic mod w on Montgomery residues - ~1.3 k rounds of up to 64 independent
// records (linear combinations / Montgomery products) instead of the 16 k rounds of the integer-chip replay.  One wave
// per instance, values in LDS slots ([slot][WW] words, fully reduced), records streamed through two LDS chunk buffers
// like h2e_replay_wave.  A record with a hint slot stores its value (still in Montgomery form) into the hint workspace.
template <class FP>
WI_INLINE Wd<FP::WW> f_ld(const u64* fv, u32 slot) {
    const u64* p = fv + (size_t)slot * FP::WW;
    Wd<FP::WW> r;
#pragma unroll
    for (int i = 0; i < FP::WW / 2; i++) {
        u64x2 q = l_ld16(p + 2 * i);
        r.v[2 * i] = q.x;
        r.v[2 * i + 1] = q.y;
    }
    return r;
}
template <class FP>
WI_INLINE void f_st(u64* fv, u32 slot, const Wd<FP::WW>& v) {
    u64* p = fv + (size_t)slot * FP::WW;
#pragma unroll
    for (int i = 0; i < FP::WW / 2; i++) l_st16(p + 2 * i, v.v[2 * i], v.v[2 * i + 1]);
}
// A < 2^12 w  ->  A mod w: quotient estimate from the top 52 bits of w in double precision, then corrections
template <class FP>
WI_INLINE Wd<FP::WW> f_reduce_small(const Wd<FP::WW + 1>& A, const Wd<FP::WW>& w, double inv_w_top) {
    constexpr int N = FP::WW, SH = FP::K - 52;
    u64 a_top = wd_shr<1, SH>(A).v[0];                       // < 2^(11 + 52)
    u64 qe = (u64)((double)a_top * inv_w_top);                // inv_w_top = 1 / (w_top + 1): never more than ~1 too small ...
    qe = qe > 0 ? qe - 1 : 0;                                 // ... and with this never too large
    Wd<N + 1> we = wd_resize<N + 1>(w);
    Wd<N + 1> r = wd_sub<N + 1>(A, wd_mul_small<N>(w, (u32)qe));
#pragma unroll
    for (int it = 0; it < 3; it++) {
        bool ge = wd_geq<N + 1>(r, we);
        r = wd_select<N + 1>(ge, wd_sub<N + 1>(r, we), r);
    }
    return wd_resize<N>(r);
}
#ifdef H2E_AB_KERNELS   // (A/B kernel: H2E_FIELD_CHAIN=lanes in -DH2E_DEBUG_HOOKS builds of the C-ABI layer)
template <class FP>
__global__ void __launch_bounds__(128) h2e_field_chain(H2EPreKernel K, const u32* __restrict__ args, const u64* __restrict__ pool,
                                                        const InstanceDesc* __restrict__ inst, u32 n_instances) {
    // Two waves per instance: wave 0 computes, wave 1 only streams the record chunks into LDS (LDS-DMA) - loads and stores
    // share one in-order counter per wave, so a computing wave that waited for its own chunk loads would also wait for every
    // hint store it has issued since (a round trip to HBM).  The waves meet at one barrier per chunk.
    constexpr int N = FP::WW;
    const u32 lane = threadIdx.x & 63u;
    const bool loader = threadIdx.x >= 64;
    const u32 instance = blockIdx.x;
    InstanceDesc d = inst[instance];
    const H2EFieldConsts* fc = &g_fc[FP::ID];
    extern __shared__ ulonglong2 f_dyn[];
    H2EVRec* rbuf = (H2EVRec*)f_dyn;                       // [2][H2E_WCHUNK]
    u64* fv = (u64*)(rbuf + 2 * H2E_WCHUNK);                // [f_slots][N]
    __builtin_amdgcn_s_setprio(3);
    const H2EVRec* recs = (const H2EVRec*)(args + K.f_recs);
    const u32 n_chunks = K.f_n_recs / H2E_WCHUNK;
    auto load_chunk = [&](u32 chunk) {
        const char* src = (const char*)(recs + (size_t)chunk * H2E_WCHUNK);
        char* dst = (char*)(rbuf + (size_t)(chunk & 1u) * H2E_WCHUNK);
#pragma unroll
        for (u32 k = 0; k < H2E_WCHUNK * 32u / 1024u; k++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k * 1024u + lane * 16u),
                                             (__attribute__((address_space(3))) void*)(dst + k * 1024u), 16, 0, 0);
    };
    if (loader) {
        load_chunk(0);
        if (n_chunks > 1) load_chunk(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // barrier 0: chunks 0 and 1 are in LDS
        for (u32 c = 1; c < n_chunks; c++) {
            __builtin_amdgcn_s_barrier();                   // barrier c: the computing wave has left chunk c - 1
            if (c + 1 < n_chunks) {
                load_chunk(c + 1);                          // ... into that buffer
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        return;
    }
    Mont<N> M = mont_w<FP>(fc);
    const Wd<N> w = M.p;
    const double inv_w_top = 1.0 / (double)(wd_shr<1, FP::K - 52>(w).v[0] + 1);
    __builtin_amdgcn_s_barrier();
    u32 cur_chunk = 0;
    u32 pos = 0;   // record index of the next round's header (wave-uniform)
    // one round; LOADS = the leading rounds that bring inputs and constants in (global loads), otherwise everything else:
    // two loops, so that the main loop's body contains no global load
    auto run_round = [&](auto loads_tag) {
        constexpr bool LOADS = decltype(loads_tag)::value;
        u32 chunk = pos / H2E_WCHUNK;
        if (chunk != cur_chunk) {   // (chunks follow each other: chunk == cur_chunk + 1)
            lds_round_barrier_workgroup();   // barrier `chunk`: the loader had this chunk in LDS before the previous barrier
            cur_chunk = chunk;
        }
        // the round's header record (word 0: count | kind << 8) sits in LDS in front of its records
        const H2E_AS_LDS u32* hp = (const H2E_AS_LDS u32*)(rbuf + (size_t)(chunk & 1u) * H2E_WCHUNK + pos % H2E_WCHUNK);
        u32 meta = __builtin_amdgcn_readfirstlane(hp[0]);
        u32 max_terms = __builtin_amdgcn_readfirstlane(hp[1]);   // the round's longest linear combination
        u32 cnt = meta & 0xffu, kind = (meta >> 8) & 0xffu;
        if (kind == 0xffu) {        // the rest of this chunk is padding
            pos = (chunk + 1) * H2E_WCHUNK;
            return false;
        }
        const u32 first = pos + 1;
        pos += 1 + cnt;
        if (lane < cnt) {
            const H2E_AS_LDS u32x4* rp = (const H2E_AS_LDS u32x4*)(rbuf + (size_t)(chunk & 1u) * H2E_WCHUNK + first % H2E_WCHUNK + lane);
            u32x4 ra = rp[0], rb = rp[1];
            u32 opc = ra.x & 0xffu, dst = ra.x >> 16, hint = ra.y;
            Wd<N> out = wd_zero<N>();
            if constexpr (LOADS) {   // values entering: inputs and pool constants
                if (opc == H2E_F_INPUT_W) out = to_mont<N>(M, (ra.x & H2E_F_FROM_HINTS) ? ws_load<N>(d.hints + (size_t)ra.z * d.ws) : g_load<N>(d.inputs + (size_t)ra.z * K.n_params));
                else if (opc == H2E_F_CONST_W) out = to_mont<N>(M, g_load<N>(pool + ra.z));
                else if (opc == H2E_F_INPUT_FE) out.v[0] = g_ld8(d.inputs + (size_t)ra.z * K.n_params);
                else if (opc == H2E_F_CONST_FE) out.v[0] = g_ld8(pool + ra.z);
            } else if (kind == 2) {           // Montgomery products
                out = mont_mul_w<FP>(f_ld<FP>(fv, ra.z), f_ld<FP>(fv, ra.w));
            } else if (kind == 0) {    // light: linear combinations, conditions, selections
                if (opc == H2E_F_LIN) {
                    u32 terms[H2E_F_MAX_TERMS] = {ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
                    Wd<N + 1> acc = wd_zero<N + 1>();
#pragma unroll
                    for (int t = 0; t < H2E_F_MAX_TERMS; t++) {
                        u32 sl = terms[t] & 0xffffu;
                        if ((u32)t < max_terms && (terms[t] >> 16) != 0u) {   // (t < max_terms is wave-uniform: whole term blocks are skipped; an unused term has coefficient 0)
                            int coef = (int)(int16_t)(terms[t] >> 16);
                            Wd<N> x = f_ld<FP>(fv, sl);
                            if (coef < 0) x = wd_sub<N>(w, x);   // -x = w - x (in (0, w]: the sum stays below 2^11 w)
                            wd_mac_small<N>(acc, x, (u32)(coef < 0 ? -coef : coef));
                        }
                    }
                    out = f_reduce_small<FP>(acc, w, inv_w_top);
                } else if (opc == H2E_F_ISZERO) {
                    out.v[0] = wd_is_zero<N>(f_ld<FP>(fv, ra.z)) ? 1 : 0;
                } else if (opc == H2E_F_NOT) {
                    out.v[0] = 1 ^ (f_ld<FP>(fv, ra.z).v[0] & 1);
                } else if (opc == H2E_F_AND || opc == H2E_F_OR || opc == H2E_F_XNOR) {
                    u64 a = f_ld<FP>(fv, ra.z).v[0] & 1, b = f_ld<FP>(fv, ra.w).v[0] & 1;
                    out.v[0] = opc == H2E_F_AND ? (a & b) : opc == H2E_F_OR ? (a | b) : (1 ^ a ^ b);
                } else if (opc == H2E_F_SELECT) {
                    bool take_a = f_ld<FP>(fv, ra.z).v[0] != 0;
                    Wd<N> a = f_ld<FP>(fv, ra.w);
                    Wd<N> b = (rb.x & 0xffffu) != 0xffffu ? f_ld<FP>(fv, rb.x & 0xffffu) : wd_zero<N>();
                    out = take_a ? a : b;
                }
            } else {                   // division: a / b, 0 for b = 0 (integer_chip.rs:524-527)
                Wd<N> a = f_ld<FP>(fv, ra.z), b = f_ld<FP>(fv, ra.w);
                Wd<N> binv = wd_inv_mod<N>(from_mont<N>(M, b), w);          // plain b^-1 (0 for 0)
                out = mont_mul_w<FP>(a, to_mont<N>(M, binv));                // a R * b^-1 R / R
            }
            if (dst != 0xffffu) f_st<FP>(fv, dst, out);
            if (hint) {
                // (conditions are raw 0 / 1: stored as the Montgomery form of that number, so that the finalize kernel's
                // conversion of the whole slot range gives 0 / 1 back)
                bool raw = opc == H2E_F_ISZERO || opc == H2E_F_NOT || opc == H2E_F_AND || opc == H2E_F_OR || opc == H2E_F_XNOR ||
                           opc == H2E_F_INPUT_FE || opc == H2E_F_CONST_FE;
                Wd<N> hv = raw ? ((out.v[0] & 1) ? M.r1 : wd_zero<N>()) : out;
                ws_store<N>(d.hints + (size_t)(hint - 1) * d.ws, hv);
            }
        }
        lds_round_barrier_wave();
        return true;
    };
    for (u32 round = 0; round < K.f_n_load_rounds;)
        if (run_round(std::true_type())) round++;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the loads are done: from here on the wave only stores to global memory
    for (u32 round = K.f_n_load_rounds; round < K.f_n_rounds;)
        if (run_round(std::false_type())) round++;
}
#endif
// ------------------------------------------------------------------------------------------------
// Digit-parallel field chain.  The lane-per-record kernel above is bound by its instruction count: one lane walks a whole
// 256 / 384-bit value through its carry chains (~600 instructions per linear combination, ~700 per Montgomery product) while
// a round rarely has more than 20 - 40 records, so most of the wave idles.  Here a record is a DPP row of 16 lanes and a lane
// owns ONE 32-bit digit of the value: column sums are per-lane 64-bit multiply-adds with no carries at all, a carry chain
// over the whole value is resolved wave-wide in a handful of instructions (generate mask from v_add_co, propagate mask from
// a compare, the ripple itself is one scalar addition: ((P + (G << 1)) ^ P) is the mask of carry-ins), digits move between
// lanes by DPP (row_shr / row_shl / row_newbcast) without touching LDS.  15 computing waves (60 records per pass) + the
// loader wave; one s_barrier per round.
//   linear combination: acc_j = beta_j + sum coef_t * x_t,j  with beta = the digits of a multiple of w that are all >= 2^44
//     (H2EFieldConsts::lin_bias), so every column stays positive whatever the signs; quotient estimate from the three top
//     columns in double precision, E_j = acc_j - q w_j + (a zero-sum bias that keeps the columns positive), ONE carry resolve
//     (DigitRow::reduce_columns).  Values stay in [0, 2 w): no conditional subtraction anywhere.
//   Montgomery product: digit-serial (one round per digit of a), the columns are kept as unnormalised 64-bit values between
//     the rounds - P = a_i b_j + T_j, Q = m w_j + P with the multiply-add's carry-out as bit 64, T_j <- hi(Q_j) + 2^32 carry_j +
//     lo(Q_j+1): nine instructions per digit and no carry propagation in the loop; one carry resolve at the end.
typedef long long i64;
#define H2E_DPP_ROW_SHL1 0x101
#define H2E_DPP_ROW_SHR1 0x111
#define H2E_DPP_ROW_BCAST(n) (0x150 + (n))
#define H2E_DP_WAVES 15u
#define H2E_DP_GROUPS (H2E_DP_WAVES * 4u)
// (H2E_DP_CHUNKS - record chunks in LDS, 16 KB each - lives in tape.h: the host compiler budgets the LDS with it)
template <int CTRL>
WI_INLINE u32 dpp_mov(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true); }
template <int CTRL>
WI_INLINE double dpp_mov_f64(double x) {
    u64 b = (u64)__double_as_longlong(x);
    u64 r = pack64(dpp_mov<CTRL>((u32)b), dpp_mov<CTRL>((u32)(b >> 32)));
    return __longlong_as_double((long long)r);
}
WI_INLINE u64 mad64_co(u32 a, u32 b, u64 c, u64& carry) {   // a b + c (mod 2^64), bit 64 of the sum as a lane mask
    u64 r;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "v"(b), "v"(c));
    return r;
}
WI_INLINE u32 mad_u32_u16(u32 a16, u32 b16, u32 c) {   // (low 16 bits of a) x (low 16 bits of b) + c
    u32 r;
    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a16), "v"(b16), "v"(c));
    return r;
}
WI_INLINE u32 sel_by_mask(u32 if0, u32 if1, u64 m) {
    u32 r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(m));
    return r;
}
template <int T, int NT, class F>
WI_INLINE void static_for(F&& f) {
    if constexpr (T < NT) {
        f(std::integral_constant<int, T>());
        static_for<T + 1, NT>(f);
    }
}
template <int D>
struct DigitRow {
    u32 j;         // this lane's digit index within its row
    u32 wj;        // digit j of w (0 above the top digit)
    u64 digits;    // mask: lanes that hold a digit (j < D)
    double inv_top;   // ~ 1 / (w's top three digits), see make() and reduce_columns()
    u64 rbias;        // this lane's column of the zero-sum bias of reduce_columns()
    // digits with the carry-out mask G of the addition that made them: the carries go in (never across a row: the lanes
    // above the top digit hold zeros)
    WI_INLINE static u32 carry(u32 d, u64 G) {
        u64 P = __builtin_amdgcn_ballot_w64(d == 0xffffffffu);
        u64 C = (P + (G << 1)) ^ P;
        return addc_co32(d, 0u, C);
    }
    // sum_j (lo_j + 2^32 hi_j) 2^(32 j)  ->  digits (lane D takes what exceeds the D digits)
    WI_INLINE static u32 normalize(u32 lo, u32 hi) {
        u32 up = dpp_mov<H2E_DPP_ROW_SHR1>(hi);
        u64 G;
        u32 d = add_co32(lo, up, G);
        return carry(d, G);
    }
    // columns of a linear combination (c_j = lo_j + 2^32 hi_j < 2^47, everything above the digits zero; below 2^15 w as a
    // number) -> the value mod w in [0, 2 w), with ONE carry resolve:
    //   quotient estimate from the three top *columns* as they are (in double precision; never above the true quotient, at
    //   most one below: inv_top is the reciprocal of w's top three digits, slightly low on purpose);
    //   E_j = c_j - q w_j + bias_j, the bias K 2^32 - K [j > 0] (- K alone in lane D, K = 2^17) keeps every column positive
    //   and sums to zero over the lanes; the digits of E are the result (lane D comes out 0, what it pushes up is dropped).
    WI_INLINE u32 reduce_columns(u32 lo, u32 hi) const {
        double cd = (double)hi * 4294967296.0 + (double)lo;
        double c1 = dpp_mov_f64<H2E_DPP_ROW_BCAST(D - 1)>(cd), c2 = dpp_mov_f64<H2E_DPP_ROW_BCAST(D - 2)>(cd), c3 = dpp_mov_f64<H2E_DPP_ROW_BCAST(D - 3)>(cd);
        double top = (c1 * 4294967296.0 + c2) * 4294967296.0 + c3;
        u32 qe = (u32)(top * inv_top);
        u64 E = pack64(lo, hi) + rbias - (u64)qe * wj;
        return sel_by_mask(0u, normalize((u32)E, (u32)(E >> 32)), digits);
    }
    // this lane's view of the field (digit j of w, masks, the reciprocal of w's top two digits - slightly low on purpose)
    WI_INLINE static DigitRow make(const H2EFieldConsts* fc, u32 lane) {
        DigitRow R;
        R.j = lane & 15u;
        const bool digit_lane = R.j < (u32)D;
        R.wj = digit_lane ? ((const H2E_AS_GLOBAL u32*)fc->w)[digit_lane ? R.j : 0u] : 0u;
        const H2E_AS_GLOBAL u32* wd = (const H2E_AS_GLOBAL u32*)fc->w;
        const double w_top3 = ((double)wd[D - 1] * 4294967296.0 + (double)wd[D - 2]) * 4294967296.0 + (double)wd[D - 3];
        R.inv_top = (1.0 / w_top3) * (1.0 - 0x1p-45);
        constexpr u64 K = 1ull << 17;
        R.rbias = R.j == 0u ? K << 32 : R.j < (u32)D ? (K << 32) - K : R.j == (u32)D ? 0ull - K : 0ull;
        R.digits = __builtin_amdgcn_ballot_w64(digit_lane);
        return R;
    }
    // a b / R mod w for digit rows a, b (zero above the top digit), R = 2^(32 D).  Values live in [0, 2 w): R > 4 w for both
    // base fields (R / w = 5.3 and 9.8), so a b < R w, the result (a b + m w) / R is below 2 w again and the chain never needs
    // the canonical representative - h2e_field_finalize makes it from the stored hint values.
    template <int I>
    WI_INLINE void mont_step(u32 a, u32 b, u32 minv32, u64& T) const {   // digits i = I .. D - 1 of a
        if constexpr (I < D) {
            // P = a_i b_j + T_j (fits 64 bits: T_j < 3 2^32); m from column 0; Q = m w_j + P with its bit 64 as the multiply-add's
            // carry-out; T'_j = hi(Q_j) + 2^32 carry_j + lo(Q_j+1).  Nine instructions: written with the carry operands spelled
            // out, because from `(u64)m * w + (u32)P` the compiler builds zero-extended register pairs with moves (16).
            u32 ai = dpp_mov<H2E_DPP_ROW_BCAST(I)>(a);
            u64 c0, cq, c2;
            u64 P = mad64_co(ai, b, T, c0);
            u32 m = dpp_mov<H2E_DPP_ROW_BCAST(0)>((u32)P * minv32);
            u64 Q = mad64_co(m, wj, P, cq);
            u32 down = dpp_mov<H2E_DPP_ROW_SHL1>((u32)Q);
            u32 t_lo = add_co32((u32)(Q >> 32), down, c2);
            u32 t_hi = addc_co32(0u, 0u, cq);
            t_hi = addc_co32(t_hi, 0u, c2);
            T = pack64(t_lo, t_hi);
            mont_step<I + 1>(a, b, minv32, T);
        }
    }
    // TWO products in one row (eight-digit fields: a value leaves half of the row's 16 lanes idle): lanes 0-7 hold the digits of
    // a, b of the first product, lanes 8-15 those of the second; wj2 = digit j mod 8 of w in every lane.  The same nine
    // instructions per digit round + two more broadcasts (every half takes a_i and m from its own lane 0 / 8 + i).  What crosses
    // the halves is harmless: lane 7 takes lo(Q) of lane 8 where the one-product form takes the zero of lane D - and lo(Q) of a
    // product's column 0 is zero by construction (that is what m is for); the columns of a result below 2 w < 2^255 leave
    // nothing above digit 7 for lane 8 to take or for a carry to run into.
    template <int I>
    WI_INLINE void mont_step2(u32 a, u32 b, u32 wj2, u32 minv32, u64& T) const {
        if constexpr (I < 8) {
            u32 ai = (u32)__builtin_amdgcn_update_dpp(0, (int)a, H2E_DPP_ROW_BCAST(I), 0xf, 0x3, true);        // banks 0-1: lanes 0-7
            ai = (u32)__builtin_amdgcn_update_dpp((int)ai, (int)a, H2E_DPP_ROW_BCAST(8 + I), 0xf, 0xc, true);   // banks 2-3: lanes 8-15
            u64 c0, cq, c2;
            u64 P = mad64_co(ai, b, T, c0);
            u32 m0 = (u32)P * minv32;
            u32 m = (u32)__builtin_amdgcn_update_dpp(0, (int)m0, H2E_DPP_ROW_BCAST(0), 0xf, 0x3, true);
            m = (u32)__builtin_amdgcn_update_dpp((int)m, (int)m0, H2E_DPP_ROW_BCAST(8), 0xf, 0xc, true);
            u64 Q = mad64_co(m, wj2, P, cq);
            u32 down = dpp_mov<H2E_DPP_ROW_SHL1>((u32)Q);
            u32 t_lo = add_co32((u32)(Q >> 32), down, c2);
            u32 t_hi = addc_co32(0u, 0u, cq);
            t_hi = addc_co32(t_hi, 0u, c2);
            T = pack64(t_lo, t_hi);
            mont_step2<I + 1>(a, b, wj2, minv32, T);
        }
    }
    WI_INLINE u32 mont_mul2(u32 a, u32 b, u32 wj2, u32 minv32) const {
        static_assert(D == 8, "two products per row: eight-digit fields only");
        u64 T = 0;
        mont_step2<0>(a, b, wj2, minv32, T);
        return normalize((u32)T, (u32)(T >> 32));
    }
    WI_INLINE u32 mont_mul(u32 a, u32 b, u32 minv32) const {
        u64 T = 0;
        mont_step<0>(a, b, minv32, T);
        return normalize((u32)T, (u32)(T >> 32));
    }
};
// 64-bit words in LDS read as WAVE-UNIFORM values (v_readfirstlane: the result is an SGPR pair, and what is computed from SGPRs is
// computed on the scalar unit) and written by every lane alike - the accessor modinv62::inv_mem works through in the digit chain
template <class T>
struct UniLds {
    H2E_AS_LDS T* p;
    struct Ref {
        H2E_AS_LDS T* q;
        WI_INLINE operator T() const {
            T v = *q;
            return (T)pack64((u32)__builtin_amdgcn_readfirstlane((u32)(u64)v), (u32)__builtin_amdgcn_readfirstlane((u32)((u64)v >> 32)));
        }
        WI_INLINE const Ref& operator=(T v) const {
            *q = v;
            return *this;
        }
    };
    WI_INLINE Ref operator[](int i) const { return Ref{p + i}; }
    WI_INLINE UniLds operator+(int k) const { return UniLds{p + k}; }
};
template <class FP>
__global__ void __launch_bounds__(1024) h2e_field_chain_digits(H2EPreKernel K, const u32* __restrict__ args, const u64* __restrict__ pool,
                                                                const InstanceDesc* __restrict__ inst, u32 n_instances) {
    constexpr int N = FP::WW, D = 2 * N;
    static_assert(D + 2 <= 16, "a value and its overflow lanes fit one DPP row");
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool loader = wave == H2E_DP_WAVES;
    const u32 instance = blockIdx.x;
    if (K.f_started != nullptr && threadIdx.x == H2E_DP_WAVES * 64u)   // (the loader wave's first lane: the rows' memory counter stays clean)
        __hip_atomic_fetch_add(K.f_started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    InstanceDesc d = inst[instance];
    const H2EFieldConsts* fc = &g_fc[FP::ID];
    extern __shared__ ulonglong2 f_dyn[];
    typedef u32 Rec[16];                                     // a record: 16 words
    Rec* rbuf = (Rec*)f_dyn;                                // [H2E_DP_CHUNKS][H2E_WCHUNK]: a ring of record chunks
    u64* fv = (u64*)(rbuf + H2E_DP_CHUNKS * H2E_WCHUNK);    // [f_slots][N]
    const Rec* recs = (const Rec*)(args + K.f_recs);
    const u32 n_chunks = K.f_n_recs / H2E_WCHUNK;
    auto header = [&](u32 pos, u32& cnt, u32& kind, u32& n_conts) {   // rows of the round, its kind, second records behind the rows
        const H2E_AS_LDS u32* hp = (const H2E_AS_LDS u32*)(rbuf + (size_t)((pos / H2E_WCHUNK) % H2E_DP_CHUNKS) * H2E_WCHUNK + pos % H2E_WCHUNK);
        u32 meta = __builtin_amdgcn_readfirstlane(hp[0]);
        n_conts = __builtin_amdgcn_readfirstlane(hp[1]);
        cnt = meta & 0xffu;
        kind = (meta >> 8) & 0xffu;
    };
    if (loader) {
        // follows the round structure (every wave meets at one barrier per round) and keeps the record stream
        // H2E_DP_CHUNKS - 1 chunks ahead: on entering chunk c - every wave has left chunk c - 1 - it loads chunk
        // c + H2E_DP_CHUNKS - 1 over it, and in front of the barrier that ends chunk c's last round it waits for chunk c + 1 only
        // (the loads complete in order: the younger ones stay in flight).  One chunk ahead (~6 rounds, 7 us) was enough for a
        // run alone, not next to another run's expansion, when a load can take longer than that.
        auto load_chunk = [&](u32 chunk) {
            const char* src = (const char*)(recs + (size_t)chunk * H2E_WCHUNK);
            char* dst = (char*)(rbuf + (size_t)(chunk % H2E_DP_CHUNKS) * H2E_WCHUNK);
#pragma unroll
            for (u32 k = 0; k < H2E_WCHUNK * 64u / 1024u; k++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k * 1024u + lane * 16u),
                                                 (__attribute__((address_space(3))) void*)(dst + k * 1024u), 16, 0, 0);
        };
        for (u32 c = 0; c < H2E_DP_CHUNKS && c < n_chunks; c++) load_chunk(c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // (every wave looks at the header behind its round BEFORE the round's barrier - a padding header sends it to the next
        // chunk there - so that after the last barrier of a chunk nobody reads that chunk's buffer again)
        u32 pos = 0, cur = 0;
        for (u32 round = 0; round < K.f_n_rounds; round++) {
            u32 chunk = pos / H2E_WCHUNK;
            if (chunk != cur) {
                cur = chunk;
                if (chunk + H2E_DP_CHUNKS - 1u < n_chunks) load_chunk(chunk + H2E_DP_CHUNKS - 1u);
            }
            u32 cnt, kind, nc;
            header(pos, cnt, kind, nc);
            if (kind == 3u) {
                // a division round: lane k inverts the divisor of the round's k-th record - (b R), in [0, 2 w) - and leaves the plain
                // inverse in the record's destination slot (nothing reads that slot before the round's rows have written it)
                const Rec* cb = rbuf + (size_t)((pos / H2E_WCHUNK) % H2E_DP_CHUNKS) * H2E_WCHUNK;
                // (the inversion's state - four numbers of NL 62-bit limbs and the modulus - lives in LDS behind the value slots and is
                // streamed through a few registers, modinv62::inv_mem: with it in registers this kernel was allocated 120 instead of 57
                // VGPRs for EVERY wave and a chain's CU had no room left for another run's expansion waves - 64 x bn256 pipelined 3.10
                // against 2.98 ms per step)
                // ... and every value it computes is WAVE-UNIFORM (the whole loader wave inverts one divisor at a time; what it reads
                // from LDS goes through v_readfirstlane), so the compiler puts the arithmetic on the scalar unit: the division steps cost
                // this kernel SGPRs, not VGPRs.
                constexpr int NL = (64 * N + 61) / 62;
                H2E_AS_LDS u64* scratch = (H2E_AS_LDS u64*)(fv + (size_t)K.f_slots * N);   // 4 NL + N + 1 words (H2E_DP_DIV_SCRATCH)
                for (u32 r = 0; r < cnt; r++) {   // (a check has ONE division: the loop runs once)
                    const H2E_AS_LDS u32* rp = (const H2E_AS_LDS u32*)(cb + (pos % H2E_WCHUNK + 1u + r));
                    const u32 dslot = __builtin_amdgcn_readfirstlane(rp[0]) >> 16, bslot = __builtin_amdgcn_readfirstlane(rp[3]);
                    H2E_AS_LDS u32* v32 = (H2E_AS_LDS u32*)fv;
                    Wd<N> bw, ww;
#pragma unroll
                    for (int i = 0; i < N; i++) {
                        bw.v[i] = pack64(__builtin_amdgcn_readfirstlane(v32[bslot * (u32)D + 2u * (u32)i]), __builtin_amdgcn_readfirstlane(v32[bslot * (u32)D + 2u * (u32)i + 1u]));
                        ww.v[i] = fc->w[i];
                        scratch[4 * NL + i] = ww.v[i];
                    }
                    scratch[4 * NL + N] = 0ull;
                    if (wd_geq<N>(bw, ww)) bw = wd_sub<N>(bw, ww);          // [0, 2 w) -> [0, w)
                    Wd<N> y;
                    UniLds<long long> st{(H2E_AS_LDS long long*)scratch};
                    modinv62::inv_mem<N, UniLds<long long>, UniLds<unsigned long long>>(bw.v, UniLds<unsigned long long>{(H2E_AS_LDS unsigned long long*)(scratch + 4 * NL)}, y.v,
                                                                                        st, st + NL, st + 2 * NL, st + 3 * NL);   // 0 for 0
                    if (dslot != 0xffffu) {
#pragma unroll
                        for (int i = 0; i < N; i++) {
                            v32[dslot * (u32)D + 2u * (u32)i] = (u32)y.v[i];
                            v32[dslot * (u32)D + 2u * (u32)i + 1u] = (u32)(y.v[i] >> 32);
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            pos += 1 + cnt + nc;
            if (pos % H2E_WCHUNK != 0) {
                u32 c2, k2, m2;
                header(pos, c2, k2, m2);
                if (k2 == 0xffu) pos = (chunk + 1) * H2E_WCHUNK;
            }
            // the chunk's last round: the next chunk must be there, the two behind it may still be on their way (16 loads each)
            static_assert(H2E_DP_CHUNKS == 4u && H2E_WCHUNK * 64u / 1024u == 16u, "the wait below counts the loads of two chunks");
            // (in the tail no younger loads were issued behind chunk + 1: the count to leave in flight shrinks with them - a
            // fixed vmcnt(32) would let the last two chunks go unwaited-for)
            if (pos / H2E_WCHUNK != chunk) {
                if (chunk + 3u < n_chunks) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                else if (chunk + 2u < n_chunks) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            lds_round_barrier_workgroup();
        }
        return;
    }
    __builtin_amdgcn_s_setprio(3);
    DigitRow<D> R = DigitRow<D>::make(fc, lane);
    const u32 j = R.j, row_base = lane & 48u;
    const bool digit_lane = j < (u32)D;
    const u32 jd = digit_lane ? j : 0u;
    const u32 r1j = digit_lane ? ((const H2E_AS_GLOBAL u32*)fc->w_r1)[jd] : 0u;
    const u32 r2j = digit_lane ? ((const H2E_AS_GLOBAL u32*)fc->w_r2)[jd] : 0u;
    const u64 beta = digit_lane ? ((const H2E_AS_GLOBAL u64*)fc->lin_bias)[jd] : 0ull;
    const u32 minv32 = (u32)fc->w_minv;
    const u32 ej = R.wj - (j == 0u ? 2u : 0u);   // digit j of w - 2 (w is odd and > 2: no borrow leaves digit 0)
    const u32 wj2 = D == 8 ? ((const H2E_AS_GLOBAL u32*)fc->w)[j & 7u] : 0u;   // digit j mod 8 of w: the second product of a paired row (eight-digit fields)
    const u32 grp = wave * 4u + (lane >> 4);
    const H2E_AS_LDS u32* fv32 = (const H2E_AS_LDS u32*)fv;
    const u32 fv_digit_addr = (u32)(size_t)fv32 + j * 4u;   // LDS address of digit j of value slot 0
    const u32 fv_digit_addr_half = (u32)(size_t)fv32 + (j & 7u) * 4u;   // ... of digit j mod 8 (eight-digit fields: two terms per instruction)
    auto ld_digit = [&](u32 slot) -> u32 {       // digit j of a value slot (whatever lies behind it for the lanes above: masked by the caller)
        return fv32[slot * (u32)D + j];
    };
    auto ld_value = [&](u32 slot) -> u32 { return sel_by_mask(0u, ld_digit(slot), R.digits); };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    u32 pos = 0, n_cnt = 0, n_kind = 0, n_nc = 0;
    bool n_valid = false;
    u32 n_rec = 0;   // this row's record of the next round: lane j holds word j (a word is broadcast over the row by DPP when it is used)
#define DP_STAMP(i, v)
    auto run_round = [&](auto loads_tag) {
        constexpr bool LOADS = decltype(loads_tag)::value;
        u32 chunk = pos / H2E_WCHUNK;
        u32 cnt = n_cnt, kind = n_kind, n_conts = n_nc;
        const Rec* cbuf = rbuf + (size_t)(chunk % H2E_DP_CHUNKS) * H2E_WCHUNK;
        auto rec_ptr = [&](u32 at) {   // record `at` of this chunk (a row beyond the round's records reads some record: unused)
            return (const H2E_AS_LDS u32*)(cbuf + (at < H2E_WCHUNK ? at : H2E_WCHUNK - 1u));
        };
        u32 r0 = n_rec;
        if (!n_valid) {   // the first round of a chunk
            header(pos, cnt, kind, n_conts);
            r0 = rec_ptr(pos % H2E_WCHUNK + 1u + grp)[j];
        }
        const u32 first = pos + 1;
        pos += 1 + cnt + n_conts;   // (the second records of the round's fused products sit behind its rows)
        // What lies behind this round is read now, before the round's barrier - the loader may overwrite this chunk's buffer
        // after the barrier of the chunk's last round - and used after the round's work, so that these LDS round trips run
        // under it: the next header (padding = on to the next chunk, whose first header is read after the barrier that its
        // data is complete at) and this row's record of the next round.
        const bool same_chunk = pos % H2E_WCHUNK != 0;
        u32 ph0 = 0, ph1 = 0;
        u32 pr = r0;
        if (same_chunk) {
            const H2E_AS_LDS u32* hp = rec_ptr(pos % H2E_WCHUNK);
            ph0 = hp[0];
            ph1 = hp[1];
            pr = rec_ptr(pos % H2E_WCHUNK + 1u + grp)[j];
        }
        if (!LOADS && kind == 3u) __builtin_amdgcn_s_barrier();   // a division round: the loader wave has left the inverses in the destination slots
        u32 rw_pass = r0;
        for (u32 op = grp; op < cnt; op += H2E_DP_GROUPS) {
            const u32 rw = rw_pass;
            // (a round of more than H2E_DP_GROUPS rows is several passes: this row's record of the next pass is read now, its LDS round
            // trip runs under this pass's work)
            if (op + H2E_DP_GROUPS < cnt) rw_pass = rec_ptr(first % H2E_WCHUNK + op + H2E_DP_GROUPS)[j];
            const u32 w0 = dpp_mov<H2E_DPP_ROW_BCAST(0)>(rw), w1 = dpp_mov<H2E_DPP_ROW_BCAST(1)>(rw);
            const u32 hint = w1 & 0x3ffffu;   // (bits 18-31: the sum of a linear combination's coefficients)
            const u32 w2 = dpp_mov<H2E_DPP_ROW_BCAST(2)>(rw), w3 = dpp_mov<H2E_DPP_ROW_BCAST(3)>(rw), w4 = dpp_mov<H2E_DPP_ROW_BCAST(4)>(rw);
            u32 opc = w0 & 0xfu, dst = w0 >> 16;   // (bits 4-7: terms of a combination, 8-15: a fused product's second record)
            u32 out = 0;
            bool raw = false;
            u32 st_dst = dst, st_hint = hint, st_j = j;   // where this LANE's digit of the result goes (a pair of products: two destinations per row)
            bool st_ok = digit_lane;
            if constexpr (LOADS) {   // values entering: inputs and pool constants
                const u64* src = (opc == H2E_F_INPUT_W || opc == H2E_F_INPUT_FE) ? d.inputs + (size_t)w2 * K.n_params : pool + w2;
                if (opc == H2E_F_INPUT_W && (w0 & H2E_F_FROM_HINTS)) src = d.hints + (size_t)w2 * d.ws;   // a value an earlier segment left in a hint slot
                bool wide = opc == H2E_F_INPUT_W || opc == H2E_F_CONST_W;
                u32 x = 0;
                if (wide ? digit_lane : j < 2u) x = ((const H2E_AS_GLOBAL u32*)src)[j];
                if (wide) out = R.mont_mul(x, r2j, minv32);
                else {
                    out = x;
                    raw = true;
                }
            } else if (kind == 0) {    // linear combinations, conditions, selections - and products (mixed rounds: the host gives every
                                       // kind its own waves)
                if (opc == H2E_F_MUL) {
                    bool packed = false;
                    if constexpr (D == 8) {
                        // a pair of products in one row (record flag in the term-count bits; the host pairs the products of a round):
                        // every row of a wave that holds a pair runs the two-product form, a single product as a pair without a second half
                        const bool pair = ((w0 >> 4) & 0xfu) == 1u;
                        if (__builtin_amdgcn_ballot_w64(pair) != 0ull) {
                            packed = true;
                            const bool up = j >= 8u;
                            const u32 w5 = dpp_mov<H2E_DPP_ROW_BCAST(5)>(rw);
                            const u32 sa = up ? (w2 >> 16) : (w2 & 0xffffu), sb = up ? (w3 >> 16) : (w3 & 0xffffu);
                            const bool live = !up || pair;
                            u32 xa = fv32[sa * (u32)D + (j & 7u)], xb = fv32[sb * (u32)D + (j & 7u)];
                            xa = live ? xa : 0u;
                            xb = live ? xb : 0u;
                            out = R.mont_mul2(xa, xb, wj2, minv32);
                            st_dst = up ? (pair ? (w4 & 0xffffu) : 0xffffu) : dst;
                            st_hint = up ? (pair ? (w5 & 0x3ffffu) : 0u) : hint;
                            st_j = j & 7u;
                            st_ok = true;
                        }
                    }
                    if (!packed) out = R.mont_mul(ld_value(w2 & 0xffffu), ld_value(w3 & 0xffffu), minv32);
                } else if (opc == H2E_F_LIN) {
                    // A linear combination: columns acc_j = beta_j + sum coef_t x_t,j.  The digits are read as
                    // x - 2^31 (one xor), which makes a term one signed multiply-add; the host put sum coef_t into the record
                    // (word 1, bits 18-31) and (sum coef_t) 2^31 goes back in at the start.
                    // A LONG combination (15 .. 28 terms, round 5) has a second record behind the round's rows (index in bits 8-15 of
                    // word 0, 0 = none): its terms go into the same columns before the one reduction - a level of the program less
                    // wherever a sum of up to 28 products would have been a partial sum and a final record.
                    auto combination = [&](u32 rwx, u32 nt, u32 v1, u32 cidx2) -> u32 {
                        i64 acc = (i64)beta + ((i64)((int)v1 >> 18) << 31);
                        if constexpr (D == 8) {
                            if (j >= 8u) acc = 0;   // (beta is zero up there already; the coefficient sum goes in once)
                        }
                        // (an unused term has coefficient 0 and slot 0; all digits are read before the first is used: one LDS
                        // round trip per combination, not one per term)
                        auto combine = [&](u32 rwy, auto nt_tag) {
                            constexpr int NT = decltype(nt_tag)::value;
                            if constexpr (D == 8) {
                                // eight digits leave half of the row idle: lanes 0-7 take the even terms, lanes 8-15 the odd ones
                                // (digit j - 8), the two half sums meet at the end - half the instructions per term
                                u32 x[NT / 2];
                                int coef[NT / 2];
                                static_for<0, NT / 2>([&](auto tc) {
                                    constexpr int P = decltype(tc)::value;
                                    u32 term = (u32)__builtin_amdgcn_update_dpp(0, (int)rwy, H2E_DPP_ROW_BCAST(2 + 2 * P), 0xf, 0x3, true);        // banks 0-1: lanes 0-7
                                    term = (u32)__builtin_amdgcn_update_dpp((int)term, (int)rwy, H2E_DPP_ROW_BCAST(3 + 2 * P), 0xf, 0xc, true);   // banks 2-3: lanes 8-15
                                    x[P] = *(const H2E_AS_LDS u32*)(size_t)mad_u32_u16(term, (u32)D * 4u, fv_digit_addr_half);
                                    coef[P] = (int)term >> 16;
                                });
                                DP_STAMP(1, x[NT / 2 - 1]);
#pragma unroll
                                for (int t = 0; t < NT / 2; t++) acc += (i64)coef[t] * (i64)(int)(x[t] ^ 0x80000000u);
                            } else {
                                u32 x[NT];
                                int coef[NT];
                                static_for<0, NT>([&](auto tc) {
                                    constexpr int T = decltype(tc)::value;
                                    u32 term = dpp_mov<H2E_DPP_ROW_BCAST(2 + T)>(rwy);
                                    x[T] = *(const H2E_AS_LDS u32*)(size_t)mad_u32_u16(term, (u32)D * 4u, fv_digit_addr);   // slot (low 16 bits) x 4 D + (slot 0's digit j)
                                    coef[T] = (int)term >> 16;
                                });
                                DP_STAMP(1, x[NT - 1]);
                                i64 acc1 = 0;   // (two chains of multiply-adds instead of one of NT)
#pragma unroll
                                for (int t = 0; t < NT; t += 2) {
                                    acc += (i64)coef[t] * (i64)(int)(x[t] ^ 0x80000000u);
                                    acc1 += (i64)coef[t + 1] * (i64)(int)(x[t + 1] ^ 0x80000000u);
                                }
                                acc += acc1;
                            }
                        };
                        // the term loop of this wave's longest combination (the host sorts a round's records by their length)
                        auto combine_n = [&](u32 rwy, u32 n) {
                            if (__builtin_amdgcn_ballot_w64(n > 10u)) combine(rwy, std::integral_constant<int, H2E_F_MAX_TERMS_WIDE>());
                            else if (__builtin_amdgcn_ballot_w64(n > 6u)) combine(rwy, std::integral_constant<int, 10>());
                            else if (__builtin_amdgcn_ballot_w64(n > 2u)) combine(rwy, std::integral_constant<int, 6>());
                            else combine(rwy, std::integral_constant<int, 2>());
                        };
                        // (the second record is read before the first one's terms are worked through: one LDS round trip less in the row's path;
                        // a row of this wave without a second record reads some record and takes none of its terms)
                        const bool any2 = __builtin_amdgcn_ballot_w64(cidx2 != 0u) != 0ull;
                        u32 rw2 = 0u;
                        if (any2) {
                            rw2 = rec_ptr(first % H2E_WCHUNK + cidx2)[j];
                            rw2 = cidx2 != 0u ? rw2 : 0u;
                        }
                        combine_n(rwx, nt);
                        if (any2) combine_n(rw2, (dpp_mov<H2E_DPP_ROW_BCAST(0)>(rw2) >> 4) & 0xfu);
                        if constexpr (D == 8) {
                            u32 up_lo = dpp_mov<0x108>((u32)(u64)acc), up_hi = dpp_mov<0x108>((u32)((u64)acc >> 32));   // row_shl:8: lane j gets lane j + 8
                            acc += (i64)pack64(up_lo, up_hi);
                        }
                        u32 lo = sel_by_mask(0u, (u32)(u64)acc, R.digits), hi = sel_by_mask(0u, (u32)((u64)acc >> 32), R.digits);
                        DP_STAMP(2, lo);
                        return R.reduce_columns(lo, hi);                     // in [0, 2 w)
                    };
                    DP_STAMP(0, rw);
                    out = combination(rw, (w0 >> 4) & 0xfu, w1, (w0 >> 8) & 0xffu);
                    DP_STAMP(6, out);
                } else if (opc == H2E_F_ISZERO) {
                    u32 x = ld_value(w2);                                    // in [0, 2 w): zero is 0 or w
                    u64 nz = __builtin_amdgcn_ballot_w64(x != 0u), nw = __builtin_amdgcn_ballot_w64(x != R.wj);
                    out = (j == 0u && (((u32)(nz >> row_base) & 0xffffu) == 0u || ((u32)(nw >> row_base) & 0xffffu) == 0u)) ? 1u : 0u;
                    raw = true;
                } else if (opc == H2E_F_NOT) {
                    out = j == 0u ? 1u ^ (ld_digit(w2) & 1u) : 0u;
                    raw = true;
                } else if (opc == H2E_F_AND || opc == H2E_F_OR || opc == H2E_F_XNOR) {
                    u32 a = ld_digit(w2) & 1u, b = ld_digit(w3) & 1u;
                    u32 v = opc == H2E_F_AND ? (a & b) : opc == H2E_F_OR ? (a | b) : (1u ^ a ^ b);
                    out = j == 0u ? v : 0u;
                    raw = true;
                } else if (opc == H2E_F_SELECT) {
                    u32 cd = ld_digit(w2);
                    u32 c0 = dpp_mov<H2E_DPP_ROW_BCAST(0)>(cd) | dpp_mov<H2E_DPP_ROW_BCAST(1)>(cd);
                    u32 a = ld_value(w3);
                    u32 b = (w4 & 0xffffu) != 0xffffu ? ld_value(w4 & 0xffffu) : 0u;
                    out = c0 != 0u ? a : b;
                }
            } else {                   // division: a / b, 0 for b = 0 (integer_chip.rs:524-527)
                // (b R)^-1 comes from the LOADER wave (below: division steps, modinv62.h, one lane per division of the round, while the
                // rows wait at the extra barrier of a division round) in the record's destination slot; (b R)^-1 R^2 by two products
                // with R^2, then a R times it.  b^(w - 2) by square and multiply in digit rows took 381 / 571 dependent products
                // (100 / 230 us: the one round of a check's final exponentiation nothing else can run beside); the inversion code in
                // the rows' own path cost them registers and spills in every round (bls12_381 chain 2.6 -> 4.1 ms) - the loader's
                // path is a different branch of the kernel.
                u32 a = ld_value(w2);
                u32 yd = ld_value(dst);                                       // (b R)^-1, plain digits
                yd = R.mont_mul(yd, r2j, minv32);                             // (b R)^-1 R
                yd = R.mont_mul(yd, r2j, minv32);                             // (b R)^-1 R^2
                out = R.mont_mul(a, yd, minv32);                              // a R (b R)^-1 R^2 / R = (a / b) R
                (void)ej;
            }
            if (st_dst != 0xffffu && st_ok) ((H2E_AS_LDS u32*)fv)[st_dst * (u32)D + st_j] = out;
            if (__builtin_amdgcn_ballot_w64(st_hint != 0u) != 0ull) {
                // (conditions are raw 0 / 1: stored as the Montgomery form of that number, so that the finalize kernel's
                // conversion of the whole slot range gives 0 / 1 back)
                u32 hv = out;
                if (raw) hv = (dpp_mov<H2E_DPP_ROW_BCAST(0)>(out) & 1u) ? r1j : 0u;
                if (st_ok && st_hint != 0u) ((H2E_AS_GLOBAL u32*)(d.hints + (size_t)(st_hint - 1) * d.ws))[st_j] = hv;
            }
        }
        n_valid = false;
        if (same_chunk) {
            u32 meta = __builtin_amdgcn_readfirstlane(ph0);
            n_nc = __builtin_amdgcn_readfirstlane(ph1);
            n_cnt = meta & 0xffu;
            n_kind = (meta >> 8) & 0xffu;
            if (n_kind == 0xffu) pos = (chunk + 1) * H2E_WCHUNK;
            else n_valid = true;
            n_rec = pr;
        }
        lds_round_barrier_workgroup();
    };
    for (u32 round = 0; round < K.f_n_load_rounds; round++) run_round(std::true_type());
    for (u32 round = K.f_n_load_rounds; round < K.f_n_rounds; round++) run_round(std::false_type());
}
// TEST HOOK: the digit-row primitives on caller-supplied rows, one 16-lane row per case (tests/test_digit_rows_gpu.py feeds
// the patterns random data never produces: runs of 0xffffffff digits under a carry, quotient estimates on the boundary).  in: [cases][2][16] words, out: [cases][16] words.
//   op 0: normalize(lo = in0, hi = in1)   2: mont_mul(in0, in1)   3: reduce_columns(lo = in0, hi = in1)
template <class FP>
__global__ void __launch_bounds__(64) h2e_digit_rows_selftest(u32 op, u32 n_cases, const u32* __restrict__ in, u32* __restrict__ out) {
    constexpr int D = 2 * FP::WW;
    const H2EFieldConsts* fc = &g_fc[FP::ID];
    DigitRow<D> R = DigitRow<D>::make(fc, threadIdx.x);
    u32 c = blockIdx.x * 4u + (threadIdx.x >> 4);
    bool live = c < n_cases;
    if (!live) c = n_cases - 1;
    u32 a = in[((size_t)c * 2 + 0) * 16 + R.j], b = in[((size_t)c * 2 + 1) * 16 + R.j];
    u32 r = 0;
    if (op == 0) r = R.normalize(a, b);
    else if (op == 2) r = R.mont_mul(a, b, (u32)fc->w_minv);
    else r = R.reduce_columns(a, b);
    if (live) out[(size_t)c * 16 + R.j] = r;
}
extern "C" int H2E_UNIT(h2e_engine_digit_rows_selftest)(int field_pair, uint32_t op, uint32_t n_cases, const void* in, void* out, hipStream_t stream) {
    if (n_cases == 0) return 0;
    dim3 grid((n_cases + 3) / 4), block(64);
    switch (field_pair) {
#if H2E_HAS_FP(0)
        case 0: hipLaunchKernelGGL(h2e_digit_rows_selftest<FP_BN256_FQ>, grid, block, 0, stream, op, n_cases, (const u32*)in, (u32*)out); break;
#endif
#if H2E_HAS_FP(1)
        case 1: hipLaunchKernelGGL(h2e_digit_rows_selftest<FP_BLS_FQ>, grid, block, 0, stream, op, n_cases, (const u32*)in, (u32*)out); break;
#endif
        default: return -1;
    }
    return (int)hipGetLastError();
}
// hint slots [first, first + n) of every instance: Montgomery form -> canonical value
template <class FP>
__global__ void __launch_bounds__(64) h2e_field_finalize(u32 first, u32 n, const InstanceDesc* inst, u32 n_instances) {
    constexpr int N = FP::WW;
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * n_instances) return;
    u32 instance = gid % n_instances, slot = first + gid / n_instances;   // instance-minor like the workspace
    InstanceDesc d = inst[instance];
    Mont<N> M = mont_w<FP>(&g_fc[FP::ID]);
    u64* p = d.hints + (size_t)slot * d.ws;
    ws_store<N>(p, from_mont<N>(M, ws_load<N>(p)));
}

// hint-only linear combinations, after the chain and its finalize kernel (field_chain.hpp "hint-only combinations leave the
// chain"): one lane per (combination, instance), terms = canonical values in hint slots / inputs / pool constants
template <class FP>
__global__ void __launch_bounds__(64) h2e_field_sinks(H2EPreKernel K, const u32* __restrict__ args, const u64* __restrict__ pool,
                                                       const InstanceDesc* __restrict__ inst, u32 n_instances) {
    constexpr int N = FP::WW;
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= K.f_n_sinks * n_instances) return;
    u32 instance = gid % n_instances, sink = gid / n_instances;   // instance-minor like the workspace
    InstanceDesc d = inst[instance];
    Mont<N> M = mont_w<FP>(&g_fc[FP::ID]);
    const Wd<N> w = M.p;
    const double inv_w_top = 1.0 / (double)(wd_shr<1, FP::K - 52>(w).v[0] + 1);
    const u32* rec = args + K.f_sink_words + args[K.f_sinks + sink];
    const u32 nt = rec[1];
    Wd<N + 1> acc = wd_zero<N + 1>();
    for (u32 t = 0; t < nt; t++) {
        u32 term = rec[2 + t];
        int coef = (int)term >> 23;   // 9 bits, signed
        u32 kind = (term >> 21) & 3u, index = term & 0x1fffffu;
        Wd<N> x = kind == 0 ? ws_load<N>(d.hints + (size_t)index * d.ws) : kind == 1 ? g_load<N>(d.inputs + (size_t)index * K.n_params) : g_load<N>(pool + index);
        if (coef < 0) x = wd_sub<N>(w, x);   // (in (0, w]: at most 14 x 255 terms, the sum stays below 2^12 w)
        wd_mac_small<N>(acc, x, (u32)(coef < 0 ? -coef : coef));
    }
    ws_store<N>(d.hints + (size_t)rec[0] * d.ws, f_reduce_small<FP>(acc, w, inv_w_top));
}

// ------------------------------------------------------------------------------------------------
// Scan predictors.  The two long chains of msm_unsafe - a window's sum over its groups (ecc_chip.rs:320-336) and the
// accumulation over the windows (ecc_chip.rs:355-362) - are sums / a Horner recurrence in the group, so a lane does
// not have to walk them from the start: chunk sums first, a short serial pass over the chunks, then every chunk's
// *real* operations (the ones whose lambda = num / den and Jacobian result the hints are made from) in parallel from
// the chunk's start value.  The start value only has to be *some* Jacobian representation of the right point: the
// records stay self-consistent (finalize_ecc divides by each record's own Z), and the hints are canonical affine
// values.  A scan-only addition can hit equal / opposite points although the real chain does not (result Z = 0,
// which every later formula propagates); the lanes that find a start value with Z = 0 walk the real chain instead.
static __device__ unsigned long long g_scan_fallbacks;   // lanes that walked the real chain (h2e_engine_scan_fallbacks)
static __device__ u32 g_scan_test;   // test knob (h2e_engine_set_tuning(3, mask)): 1 = treat odd window chunks' offsets as degenerate,
                                     // 2 = odd tail chunks' sums, 4 = the tail's in-chunk start values of odd windows
template <class FP>
WI_INLINE void vc_init(VC& v, const InstanceDesc& d, u32 n_instances, const u32* params, const u32* aux, const H2EFieldConsts* fc, u32 lane) {
    v.c.base = d.base;
    v.c.range = d.range;
    v.c.select = d.select;
    v.c.inputs = d.inputs;
    v.c.status = d.status;
    v.c.ob = v.c.orr = v.c.os = 0;
    v.c.hs = d.hs;
    v.c.params = params;
    v.c.aux = aux;
    v.c.pool = nullptr;
    v.c.fc = fc;
    v.c.strand = lane;
    v.c.input_stride = 0;
    v.c.hints = nullptr;
    v.c.ws = d.ws;
    v.c.hint_stride = 0;
    v.c.ws = d.ws;
    v.nd = d.nd;
    v.jac = d.jac;
}
// phase 0: chunk sums S_c (lane = window x chunk); 1: offsets O_0 = -r1, O_(c+1) = S_c + O_c (lane = window);
// 2: the chunk's additions acc = C_g[idx] + acc from O_c, with their records (lane = window x chunk)
template <class FP>
__global__ void __launch_bounds__(64, H2E_CHAIN_WAVES) h2e_predict_windows(H2EPreKernel K, u32 phase, const u32* args, const InstanceDesc* inst, u32 n_instances,
                                                          const H2EFieldConsts* fc) {
    constexpr int L = FP::L, NW = FP::WW, NR = 2 * (L + 1);
    constexpr u32 G = H2E_WIN_CHUNKS;
    __builtin_amdgcn_s_setprio(3);
    const u32* a = args + K.args_begin;
    u32 n_groups = a[0];
    const u32* neg_r1 = a + 3;
    u32 len = (n_groups + G - 1) / G, nch = (n_groups + len - 1) / len;
    u32 per_instance = phase == 1 ? K.n_lanes : K.n_lanes * nch;
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_instances * per_instance) return;
    u32 instance = gid % n_instances, rest = gid / n_instances;
    u32 window = phase == 1 ? rest : rest / nch, ch = phase == 1 ? 0 : rest % nch;
    InstanceDesc d = inst[instance];
    VC v;
    vc_init<FP>(v, d, n_instances, nullptr, nullptr, fc, window);
    MontW<FP> M;
    (Mont<NW>&)M = mont_w<FP>(fc);
    u32 s0 = K.scan_begin + window * 2 * G;   // S_c at s0 + c, O_c at s0 + G + c
    // the candidate of group g in Montgomery form (x, y): value slots 2, 3 of its selection-buffer entry
    const size_t sel_step = (size_t)H2E_SEL_SLOTS * d.ws;
    const u64* selp = d.sel + (K.sel_begin + window * n_groups) * sel_step + 2 * (size_t)d.ws;
    auto neg_r1_point = [&]() {
        Jac<NW> p;
        p.x = ld_w_mont<FP>(v.c, M, neg_r1);
        p.y = ld_w_mont<FP>(v.c, M, neg_r1 + L + 1);
        p.z = M.r1;
        return p;
    };
    // acc = C_g + acc over groups [g0, g1), the candidate fetched one iteration ahead; REC: leave the ops' records
    auto walk = [&](Jac<NW> acc, u32 g0, u32 g1, bool rec, u32 hint0) {
        if (g0 >= g1) return acc;
        const u64* np = selp + g0 * sel_step;
        Wd<NW> nx = ws_load<NW>(np), ny = ws_load<NW>(np + d.ws);
        for (u32 g = g0; g < g1; g++) {
            Wd<NW> sx = nx, sy = ny;
            np = selp + min(g + 1, g1 - 1) * sel_step;
            nx = ws_load<NW>(np);
            ny = ws_load<NW>(np + d.ws);
            asm volatile("" ::: "memory");
            Wd<NW> num;
            acc = jac_madd(M, acc, sx, sy, num);
            if (rec) st_rec<FP>(v, hint0 + H2E_ECC_HINT_SLOTS * g, num, acc.z, acc.x, acc.y);
        }
        return acc;
    };
    u32 g0 = ch * len, g1 = min(g0 + len, n_groups);
    // The candidates of consecutive groups carry opposite blinding points (+-r2) and a group whose bits are all zero
    // contributes nothing else: a bare sum of candidates runs into P + (-P) all the time (two all-zero groups in a row:
    // 2^-10 per pair).  The chunk sums therefore start from -2 r1 - as safe as the real chain, which starts from -r1 -
    // and the serial pass takes the 2 r1 out again: S'_c = -2 r1 + sum, O_(c+1) = (O_c + S'_c) + 2 r1.
    if (phase == 0) {
        if (ch + 1 == nch) return;   // nothing reads the last chunk's sum
        Wd<NW> num;
        Jac<NW> s = jac_dbl(M, neg_r1_point(), num);
        s = walk(s, g0, g1, false, 0);
        st_jac<FP>(v, s0 + ch, s);
    } else if (phase == 1) {
        Wd<NW> num;
        Jac<NW> o = neg_r1_point();
        Jac<NW> two_r1 = jac_dbl(M, o, num);
        two_r1.y = mont_sub<NW>(M, wd_zero<NW>(), two_r1.y);
        for (u32 c = 0; c + 1 < nch; c++) {
            o = jac_add(M, ld_jac<FP>(v, s0 + c), o, num);
            o = jac_add(M, two_r1, o, num);
            st_jac<FP>(v, s0 + G + c + 1, o);
        }
    } else {
        u32 hint0 = K.hint_base + window * K.hints_per_lane;
        Jac<NW> acc;
        if (ch == 0) {
            acc = neg_r1_point();
            st_rec<FP>(v, hint0 + H2E_ECC_HINT_SLOTS * n_groups, acc.x, acc.z, acc.x, acc.y);   // the chain's initial point
        } else {
            acc = ld_jac<FP>(v, s0 + G + ch);
            if (wd_is_zero<NW>(acc.z) || ((g_scan_test & 1u) && (ch & 1u))) {
                atomicAdd(&g_scan_fallbacks, 1ull);
                acc = walk(neg_r1_point(), 0, g0, false, 0);
            }
        }
        acc = walk(acc, g0, g1, !(g_scan_test & 8u), hint0);   // (8: timing experiment without the records)
        if (ch == nch - 1) st_jac<FP>(v, K.scratch_begin + window, acc);
    }
}

// The tail: acc = r1; per window w: acc = 2 acc; acc = line_w + acc; [acc = acc + (-r2)], i.e. acc_w = 2 acc_(w-1) + T_w.
// phase 0: local Horner sums B_w of a chunk (lane = chunk); 1: per chunk the doubling chain D_w = 2^(j+1) A_(c-1) and
// A_c = D + B at the chunk's end (lane = instance; the only serial part: one doubling per window);
// 2: window w starts from acc_(w-1) = D_(w-1) + B_(w-1) and does its real operations, with their records (lane = window)
template <class FP>
__global__ void __launch_bounds__(64, H2E_CHAIN_WAVES) h2e_predict_tail(H2EPreKernel K, u32 phase, const u32* args, const InstanceDesc* inst, u32 n_instances,
                                                       const H2EFieldConsts* fc) {
    constexpr int L = FP::L, NW = FP::WW, NR = 2 * (L + 1);
    constexpr u32 CH = H2E_TAIL_CHUNK;
    __builtin_amdgcn_s_setprio(3);
    const u32* a = args + K.args_begin;
    u32 windows = a[0], odd = a[1];
    const u32* r1 = a + 2;
    const u32* neg_r2 = a + 2 + NR;
    u32 line0 = a[2 + 2 * NR];
    u32 nch = (windows + CH - 1) / CH;
    u32 per_instance = phase == 0 ? nch : (phase == 1 || phase == 3) ? 1 : windows;
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_instances * per_instance) return;
    u32 instance = gid % n_instances, rest = gid / n_instances;
    InstanceDesc d = inst[instance];
    VC v;
    vc_init<FP>(v, d, n_instances, nullptr, nullptr, fc, 0);
    MontW<FP> M;
    (Mont<NW>&)M = mont_w<FP>(fc);
    u32 sB = K.scan_begin, sD = sB + windows, sA = sD + windows, sX = sA + nch;   // sX: (r1.x, r1.y in Montgomery form, fallback flag) for the digit-row form of phase 1
    if (phase == 3) {
        if (((const u32*)(d.jac + ((size_t)sX * 3 + 2) * d.ws))[0] == 0u) return;   // (h2e_predict_tail_rows met no degenerate sum for this instance)
        atomicAdd(&g_scan_fallbacks, 1ull);                                       // an instance redone the lane way is a fallback
    }
    Wd<NW> bx = ld_w_mont<FP>(v.c, M, neg_r2), by = ld_w_mont<FP>(v.c, M, neg_r2 + L + 1);
    auto r1_point = [&]() {
        Jac<NW> p;
        p.x = ld_w_mont<FP>(v.c, M, r1);
        p.y = ld_w_mont<FP>(v.c, M, r1 + L + 1);
        p.z = M.r1;
        return p;
    };
    // the real operations of windows [w0, w1) from acc; REC: leave their records
    auto walk = [&](Jac<NW> acc, u32 w0, u32 w1, bool rec) {
        if (w0 >= w1) return acc;
        u32 h = K.hint_base + H2E_ECC_HINT_SLOTS * (2 + odd) * w0;
        Jac<NW> next = ld_jac<FP>(v, line0 + w0);
        for (u32 w = w0; w < w1; w++) {
            Jac<NW> line = next;
            next = ld_jac<FP>(v, line0 + min(w + 1, w1 - 1));
            asm volatile("" ::: "memory");
            Wd<NW> num;
            acc = jac_dbl(M, acc, num);
            if (rec) st_rec<FP>(v, h, num, acc.z, acc.x, acc.y);
            h += H2E_ECC_HINT_SLOTS;
            acc = jac_add(M, line, acc, num);
            if (rec) st_rec<FP>(v, h, num, acc.z, acc.x, acc.y);
            h += H2E_ECC_HINT_SLOTS;
            if (odd) {
                acc = jac_madd(M, acc, bx, by, num);
                if (rec) st_rec<FP>(v, h, num, acc.z, acc.x, acc.y);
                h += H2E_ECC_HINT_SLOTS;
            }
        }
        return acc;
    };
    auto chunk_start = [&](u32 c) { return c == 0 ? r1_point() : ld_jac<FP>(v, sA + c - 1); };
    if (phase == 0) {
        if (rest == 0) {   // the chain's start point for the digit-row kernel, and its fallback flag cleared
            Jac<NW> r = r1_point();
            r.z = wd_zero<NW>();
            st_jac<FP>(v, sX, r);
        }
        u32 w0 = rest * CH, w1 = min(w0 + CH, windows);
        Jac<NW> b;
        for (u32 w = w0; w < w1; w++) {
            Wd<NW> num;
            Jac<NW> t = ld_jac<FP>(v, line0 + w);
            if (odd) t = jac_madd(M, t, bx, by, num);
            b = w == w0 ? t : jac_add(M, t, jac_dbl(M, b, num), num);
            st_jac<FP>(v, sB + w, b);
        }
    } else if (phase == 1 || phase == 3) {
        Jac<NW> acc = r1_point();
        for (u32 c = 0; c < nch; c++) {
            u32 w0 = c * CH, w1 = min(w0 + CH, windows);
            Jac<NW> dd = acc;
            Wd<NW> num;
            for (u32 w = w0; w < w1; w++) {
                dd = jac_dbl(M, dd, num);
                st_jac<FP>(v, sD + w, dd);
            }
            Jac<NW> nx = jac_add(M, ld_jac<FP>(v, sB + w1 - 1), dd, num);
            if (wd_is_zero<NW>(nx.z) || ((g_scan_test & 2u) && (c & 1u))) {
                atomicAdd(&g_scan_fallbacks, 1ull);
                nx = walk(acc, w0, w1, false);
            }
            acc = nx;
            st_jac<FP>(v, sA + c, acc);
        }
    } else {
        u32 w = rest, c = w / CH, w0 = c * CH;
        Jac<NW> acc;
        if (w == w0) {
            acc = chunk_start(c);
        } else {
            Wd<NW> num;
            acc = jac_add(M, ld_jac<FP>(v, sB + w - 1), ld_jac<FP>(v, sD + w - 1), num);
            if (wd_is_zero<NW>(acc.z) || ((g_scan_test & 4u) && (w & 1u))) {
                atomicAdd(&g_scan_fallbacks, 1ull);
                acc = walk(chunk_start(c), w0, w, false);
            }
        }
        if (w == 0) st_rec<FP>(v, K.hint_base + H2E_ECC_HINT_SLOTS * K.ecc_ops, acc.x, acc.z, acc.x, acc.y);   // the chain's initial point
        walk(acc, w, w + 1, true);
    }
}

// Select pre-kernel of the MSM windows: one lane per (instance, window, group) does pick_candidate_non_zero
// (ecc_chip.rs:935-953) natively - index from the window's bit cells, candidate through the group's table - and leaves
// the point (canonical x, y) in the selection buffer.  The serial kernels of the value chain (windows predictor,
// replay) then only read static addresses.  Arguments as for H2E_PRE_MSM_WINDOWS.
template <class FP>
__global__ void __launch_bounds__(64) h2e_select(H2EPreKernel K, const u32* args, const u32* params_all, const u32* aux,
                                                 const InstanceDesc* inst, u32 n_instances, const H2EFieldConsts* fc) {
    constexpr int L = FP::L, NW = FP::WW, NR = 2 * (L + 1);
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_instances * K.n_lanes) return;
    u32 instance = gid % n_instances, lane = gid / n_instances;
    const u32* a = args + K.args_begin;
    u32 n_groups = a[0], group_size = a[1], n_points = a[2];
    const u32* tables = a + 3 + NR;
    u32 w = lane / n_groups, g = lane % n_groups;
    InstanceDesc d = inst[instance];
    LC c;
    c.base = d.base;
    c.range = d.range;
    c.select = d.select;
    c.ob = c.orr = c.os = 0;
    c.hs = inst[0].hs;
    c.params = params_all + K.params_begin + (size_t)w * K.n_params;
    u32 lo = g * group_size, hi = min(n_points, lo + group_size), idx = 0;
    for (u32 j = lo; j < hi; j++) idx |= (u32)(ld_limb(c, H2E_MAKE_REF(H2E_REGION_PARAM, 0, 0, j)).v[0] & 1) << (j - lo);
    const u32* tab = aux + tables[g] + idx * NR;
    u64* out = d.sel + (size_t)H2E_SEL_SLOTS * (K.sel_begin + w * n_groups + g) * d.ws;
    Mont<NW> M = mont_w<FP>(fc);
#pragma unroll
    for (int which = 0; which < 2; which++) {
        Limb l[L];
#pragma unroll
        for (int i = 0; i < L; i++) l[i] = ld_limb(c, tab[which * (L + 1) + i]);
        Wd<NW> v = wd_resize<NW>(compose<FP, FPX<FP>::AW>(l));
        ws_store<NW>(out + (size_t)which * d.ws, v);                          // canonical: the replay's / expansion's copy
        ws_store<NW>(out + (size_t)(2 + which) * d.ws, to_mont<NW>(M, v));    // Montgomery form: the scan predictor's
    }
}

// lambda = num / den for a run of hint slots: Montgomery's trick over HINT_K consecutive slots per lane, the
// hint cells hold the prefix products between the two passes; output canonical (what assign_w(c) expects).
#ifndef H2E_HINT_K
#define H2E_HINT_K 32
#endif
static constexpr int HINT_K = H2E_HINT_K;
template <class FP>
__global__ void __launch_bounds__(64, H2E_CHAIN_WAVES) h2e_finalize_hints(u32 hint_base, u32 n_hints, const InstanceDesc* inst,
                                                         u32 n_instances, const H2EFieldConsts* fc) {
    constexpr int NW = FP::WW;
    u32 chunks = (n_hints + HINT_K - 1) / HINT_K;
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_instances * chunks) return;
    u32 instance = gid % n_instances, chunk = gid / n_instances;
    InstanceDesc d = inst[instance];
    Mont<NW> M = mont_w<FP>(fc);
    u32 lo = hint_base + chunk * HINT_K, hi = min(lo + HINT_K, hint_base + n_hints);
    Wd<NW> acc = M.r1;
    for (u32 s = lo; s < hi; s++) {
        Wd<NW> den = ws_load<NW>(d.nd + ((size_t)s * 2 + 1) * d.ws);
        ws_store<NW>(d.hints + (size_t)s * d.ws, acc);
        if (!wd_is_zero<NW>(den)) acc = mont_mul<NW>(M, acc, den);
    }
    // the running inverse is kept as a plain value (not in Montgomery form): multiplied with Montgomery-form prefixes /
    // denominators / numerators it stays plain, and num * (1 / den) comes out canonical without a conversion
    Wd<NW> ainv = wd_inv_mod<NW>(from_mont<NW>(M, acc), M.p);
    for (u32 s = hi; s-- > lo;) {
        const u64* np = d.nd + (size_t)s * 2 * d.ws;
        Wd<NW> num = ws_load<NW>(np), den = ws_load<NW>(np + d.ws);
        u64* hp = d.hints + (size_t)s * d.ws;
        Wd<NW> out = wd_zero<NW>();
        if (!wd_is_zero<NW>(den)) {
            Wd<NW> dinv = mont_mul<NW>(M, ainv, ws_load<NW>(hp));
            ainv = mont_mul<NW>(M, ainv, den);
            out = mont_mul<NW>(M, num, dinv);
        }
        ws_store<NW>(hp, out);
    }
}

// Full value hints of an MSM chain (tape.h): from the records the predictor left per ecc op - numerator of lambda
// and the Jacobian result (X, Y, Z), Z being lambda's denominator - one batch inversion per ECC_CH ops gives every
// affine intermediate point, and from those the canonical value of every mul-like result of the op.  Parallel over
// (instance, chain, chunk): the chain itself was only walked by the predictor.
#ifndef H2E_ECC_CH
#define H2E_ECC_CH 32
#endif
static constexpr int ECC_CH = H2E_ECC_CH;
template <class FP>
__global__ void __launch_bounds__(64, H2E_CHAIN_WAVES) h2e_finalize_ecc(H2EPreKernel K, const InstanceDesc* inst, u32 n_instances,
                                                       const H2EFieldConsts* fc) {
    constexpr int NW = FP::WW;
    u32 chunks = (K.ecc_ops + ECC_CH - 1) / ECC_CH;
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_instances * K.n_lanes * chunks) return;
    u32 instance = gid % n_instances, chunk = (gid / n_instances) % chunks, lane = gid / (n_instances * chunks);
    InstanceDesc d = inst[instance];
    MontW<FP> M;
    (Mont<NW>&)M = mont_w<FP>(fc);
    u32 hint0 = K.hint_base + lane * K.hints_per_lane;
    int lo = (int)(chunk * ECC_CH), hi = min(lo + ECC_CH, (int)K.ecc_ops);
    // element e in [lo - 1, hi): e = -1 is the chain's initial point (block ecc_ops)
    const size_t ws = d.ws;
    auto rec = [&](int e) -> u64* { return d.nd + (size_t)(hint0 + H2E_ECC_HINT_SLOTS * (e < 0 ? K.ecc_ops : (u32)e)) * 2 * ws; };
    auto hint = [&](int e, u32 k) -> u64* { return d.hints + (size_t)(hint0 + H2E_ECC_HINT_SLOTS * (u32)e + k) * ws; };
    Wd<NW> acc = M.r1;
    for (int e = lo - 1; e < hi; e++) {
        u64* r = rec(e);
        Wd<NW> den = ws_load<NW>(r + ws);
        // prefix product -> third pair of the block's nd area (element lo - 1 also belongs to the previous chunk's
        // lane, which keeps its own prefix in the first half of that pair)
        ws_store<NW>(r + (e == lo - 1 ? 5 : 4) * ws, acc);
        if (!wd_is_zero<NW>(den)) acc = mm(M, acc, den);
    }
    // Backward pass in the *plain* domain: the running inverse is a plain value (not in Montgomery form), and a product
    // of a plain and a Montgomery-form factor is plain - so the affine coordinates, lambda and the derived values come
    // out canonical as they are, without one conversion per stored hint.  Only lambda and 1 / Z are also needed in
    // Montgomery form (as the second factor of lambda^2, (x_a - x_c) lambda, 1 / Z^2).  10 multiplications per op (was 14).
    Wd<NW> ainv = wd_inv_mod<NW>(from_mont<NW>(M, acc), M.p);
    const bool need_y = ((K.used_slots >> H2E_HINT_YC) | (K.used_slots >> H2E_HINT_AUX1)) & 1u;
    Wd<NW> xn = wd_zero<NW>(), yn = xn, ln = xn, ln_m = xn;   // op e + 1: affine result and lambda (plain; lambda also Montgomery)
    for (int e = hi - 1; e >= lo - 1; e--) {
        const u64* r = rec(e);
        Wd<NW> num = ws_load<NW>(r), den = ws_load<NW>(r + ws);
        Wd<NW> X = ws_load<NW>(r + 2 * ws), Y = need_y ? ws_load<NW>(r + 3 * ws) : wd_zero<NW>();
        Wd<NW> dinv = wd_zero<NW>();
        if (!wd_is_zero<NW>(den)) {
            dinv = mm(M, ainv, ws_load<NW>(r + (e == lo - 1 ? 5 : 4) * ws));
            ainv = mm(M, ainv, den);
        }
        Wd<NW> dinv_m = mm(M, dinv, M.r2);
        Wd<NW> zi2 = mm(M, dinv, dinv_m);
        Wd<NW> xe = mm(M, X, zi2), ye = need_y ? mm(M, Y, mm(M, zi2, dinv_m)) : wd_zero<NW>();
        Wd<NW> le = mm(M, num, dinv), le_m = mm(M, num, dinv_m);
        if (e < hi - 1) {
            // op k = e + 1: c = (xn, yn), lambda = ln, prev = (xe, ye)
            u32 k = (u32)(e + 1);
            u32 kind = (K.pattern >> (2 * (k % K.pattern_len))) & 3u;
            Wd<NW> l2 = mm(M, ln, ln_m);
            Wd<NW> xa, aux0 = wd_zero<NW>(), aux1 = wd_zero<NW>();
            if (kind == H2E_ECC_ADD_EXT_PREV) {
                xa = mont_sub<NW>(M, mont_sub<NW>(M, l2, xn), xe);      // x_a = lambda^2 - x_c - x_b
                aux0 = mont_sub<NW>(M, xa, xe);                        // x_a - x_b
            } else if (kind == H2E_ECC_ADD_PREV_EXT) {
                xa = xe;
                Wd<NW> xb = mont_sub<NW>(M, mont_sub<NW>(M, l2, xn), xa);
                aux0 = mont_sub<NW>(M, xa, xb);
            } else {
                xa = xe;
                if ((K.used_slots >> H2E_HINT_AUX0) & 1u) aux0 = mm(M, xa, mm(M, xa, M.r2));   // x_a^2
                aux1 = mont_dbl<NW>(M, ye);                             // 2 y_a
            }
            Wd<NW> t2 = mont_sub<NW>(M, xa, xn);
            Wd<NW> t2l = mm(M, t2, ln_m);
            auto put = [&](u32 slot, const Wd<NW>& v) {
                if ((K.used_slots >> slot) & 1u) ws_store<NW>(hint((int)k, slot), v);   // (a slot nobody reads is not written)
            };
            put(H2E_HINT_LAMBDA, ln);
            put(H2E_HINT_LAMBDA2, l2);
            put(H2E_HINT_XC, xn);
            put(H2E_HINT_YC, yn);
            put(H2E_HINT_T2L, t2l);
            put(H2E_HINT_T2, t2);
            put(H2E_HINT_AUX0, aux0);
            put(H2E_HINT_AUX1, aux1);
        }
        xn = xe;
        yn = ye;
        ln = le;
        ln_m = le_m;
    }
}

#if H2E_COMMON_UNIT   // kernels that do not depend on the field pair: unit 0 only
// ------------------------------------------------------------------------------------------------
// Gate in front of an expansion that becomes ready at the same moment as the next stage's digit chain (a pairing check's Miller-loop
// expansion and its final-exponentiation chain both wait for the same hint store).  A chain workgroup is 15 waves and ~100 KB of
// LDS on ONE CU; once the expansion's waves hold the SIMDs, the slots they free one by one are refilled from the expansion's own
// grid and the chain only gets its CUs when that grid has drained - a single batch of 64 bn256 checks then takes 5.5 instead of
// 4.2 ms, decided by which queue the dispatcher looked at first (1-2 us apart in profiles/r4_s_pairing_bn256_ring1).  The gate is one
// wave that waits until the chain's workgroups have reported in (H2EPreKernel::f_started) - or 30 us, so that it can never hang.
__global__ void __launch_bounds__(64) h2e_gate(const u32* counter, u32 target, u32 timeout_ticks) {
    if (threadIdx.x != 0) return;
    const u64 t0 = wall_clock64();   // 100 MHz
    while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0 && wall_clock64() - t0 < timeout_ticks)
        __builtin_amdgcn_s_sleep(8);
}
extern "C" int h2e_engine_gate(const uint32_t* counter, uint32_t target, hipStream_t stream) {
    hipLaunchKernelGGL(h2e_gate, dim3(1), dim3(64), 0, stream, counter, target, 3000u);
    return (int)hipGetLastError();
}
// ------------------------------------------------------------------------------------------------
// Hand-off (device half).  The batch-interleaved advice array [row][COLS][half][instance] of a run ->
// one array per instance in the consumer's layout: row-major [instance][row][COLS][4 words] (the reference's
// `Vec<[(Option<N>, bool); COLS]>`, src/context.rs:243-251) or column-major [instance][COLS][row][4 words] (halo2's
// advice columns; context.rs:310-541 builds them cell by cell on the host).  Cells the shape leaves unassigned come
// out as zero (`flags` = the program's assigned / permute bytes; without flags the input is copied as it is), and
// `mont` = 1 emits Montgomery-form cells (x * 2^256 mod n: the in-memory form of halo2's Fr, so the host side needs no
// per-cell `Fr::from_repr`, src/utils.rs:10-17).  A block moves TR rows x TI instances through LDS: contiguous
// TI x 16 bytes on the way in, TR x COLS x 32 (rows) or TR x 32 (columns) bytes on the way out.  HBM bound.
template <int COLS, bool COLUMNS>
__global__ void __launch_bounds__(256) h2e_export(const ulonglong2* __restrict__ in, ulonglong2* __restrict__ out,
                                                  const uint8_t* __restrict__ flags, u64 rows, u32 n_inst, u32 mont,
                                                  const H2EFieldConsts* fc) {
    constexpr int TR = 8, TI = 32, PER_I = TR * COLS * 2, PITCH = PER_I + 1;
    __shared__ ulonglong2 tile[TI * PITCH];
    u64 row0 = (u64)blockIdx.x * TR;
    u32 inst0 = blockIdx.y * TI;
    u32 nr = (u32)min((u64)TR, rows - row0), ni = min((u32)TI, n_inst - inst0);
    __shared__ uint8_t assigned[TR * COLS];   // the block's assigned flags, once; unassigned cells are not read (they come out as zero)
    if (threadIdx.x < TR * COLS) assigned[threadIdx.x] = (flags == nullptr || threadIdx.x >= nr * COLS) ? 1 : (flags[row0 * COLS + threadIdx.x] & 1);
    __syncthreads();
    for (u32 p = threadIdx.x; p < (u32)PER_I * TI; p += 256) {
        u32 i = p % TI, q = p / TI;   // q = (r * COLS + col) * 2 + half: memory order of the batch array
        if (i < ni && q < nr * COLS * 2 && assigned[q >> 1]) tile[i * PITCH + q] = in[(row0 * COLS * 2 + q) * n_inst + inst0 + i];
    }
    __syncthreads();
    Mont<4> M = mont_n(fc);
    for (u32 p = threadIdx.x; p < (u32)TR * COLS * TI; p += 256) {
        u32 i = p / (TR * COLS), k = p % (TR * COLS);
        u32 r, col;
        if (COLUMNS) { col = k / TR; r = k % TR; } else { r = k / COLS; col = k % COLS; }
        if (i >= ni || r >= nr) continue;
        u32 q = (r * COLS + col) * 2;
        ulonglong2 lo = tile[i * PITCH + q], hi = tile[i * PITCH + q + 1];
        if (!assigned[r * COLS + col]) lo = hi = make_ulonglong2(0, 0);
        if (mont) {
            Fe x;
            x.v[0] = lo.x; x.v[1] = lo.y; x.v[2] = hi.x; x.v[3] = hi.y;
            x = mont_mul<4>(M, x, M.r2);
            lo = make_ulonglong2(x.v[0], x.v[1]);
            hi = make_ulonglong2(x.v[2], x.v[3]);
        }
        u64 cell = COLUMNS ? ((u64)(inst0 + i) * COLS + col) * rows + row0 + r : ((u64)(inst0 + i) * rows + row0 + r) * COLS + col;
        out[cell * 2] = lo;
        out[cell * 2 + 1] = hi;
    }
}
// Column-major output wants long runs per (instance, column): a block takes 32 rows of ONE column of 32 instances
// through LDS - 512 contiguous bytes per (row, half) on the way in, 1 KB per (instance, column) on the way out (the
// generic tile above leaves 256-byte runs in this layout: 3.3 instead of 4.1 TB/s on the base array).
// Cells the shape leaves unassigned (a third of an MSM tile's positions) are not read - they come out as zero whatever the array holds.
// They ARE written: leaving them out (arrays zeroed once) was measured and is slower, 54.7 against 47 ms for 64 tiles - partial lines
// cost the memory system more than the zeros do.
template <int COLS>
__global__ void __launch_bounds__(256) h2e_export_columns(const ulonglong2* __restrict__ in, ulonglong2* __restrict__ out,
                                                          const uint8_t* __restrict__ flags, u64 rows, u32 n_inst, u32 mont,
                                                          const H2EFieldConsts* fc) {
    constexpr int TR = 32, TI = 32, PER_I = TR * 2, PITCH = PER_I + 1;
    __shared__ ulonglong2 tile[TI * PITCH];
    u64 row0 = (u64)blockIdx.x * TR;
    u32 inst0 = blockIdx.y * TI, col = blockIdx.z;
    u32 nr = (u32)min((u64)TR, rows - row0), ni = min((u32)TI, n_inst - inst0);
    // the block's assigned flags, once (a flag load in front of every cell load would be a dependent load per cell)
    __shared__ uint8_t assigned[TR];
    if (threadIdx.x < TR) assigned[threadIdx.x] = (flags == nullptr || threadIdx.x >= nr) ? 1 : (flags[(row0 + threadIdx.x) * COLS + col] & 1);
    __syncthreads();
    for (u32 p = threadIdx.x; p < (u32)PER_I * TI; p += 256) {
        u32 i = p % TI, q = p / TI;   // q = r * 2 + half
        if (i < ni && q < nr * 2 && assigned[q >> 1]) tile[i * PITCH + q] = in[(((row0 + (q >> 1)) * COLS + col) * 2 + (q & 1)) * n_inst + inst0 + i];
    }
    __syncthreads();
    Mont<4> M = mont_n(fc);
    for (u32 p = threadIdx.x; p < (u32)TR * TI; p += 256) {
        u32 i = p / TR, r = p % TR;
        if (i >= ni || r >= nr) continue;
        ulonglong2 lo = tile[i * PITCH + 2 * r], hi = tile[i * PITCH + 2 * r + 1];
        if (!assigned[r]) lo = hi = make_ulonglong2(0, 0);
        if (mont) {
            Fe x;
            x.v[0] = lo.x; x.v[1] = lo.y; x.v[2] = hi.x; x.v[3] = hi.y;
            x = mont_mul<4>(M, x, M.r2);
            lo = make_ulonglong2(x.v[0], x.v[1]);
            hi = make_ulonglong2(x.v[2], x.v[3]);
        }
        u64 cell = ((u64)(inst0 + i) * COLS + col) * rows + row0 + r;
        out[cell * 2] = lo;
        out[cell * 2 + 1] = hi;
    }
}
extern "C" int h2e_engine_export(uint32_t cols, int columns, int mont, const void* in, void* out, const uint8_t* flags, uint64_t rows,
                                 uint32_t n_instances, const H2EFieldConsts* fc_dev, hipStream_t stream) {
    if (rows == 0 || n_instances == 0) return 0;
    u64 tiles = (rows + 7) / 8;
    if (tiles > 0x7fffffffull) return -1;
    dim3 grid((u32)tiles, (n_instances + 31) / 32), block(256);
    if (columns) {
        dim3 gridc((u32)((rows + 31) / 32), (n_instances + 31) / 32, cols);
        switch (cols) {
            case 5: hipLaunchKernelGGL(h2e_export_columns<5>, gridc, block, 0, stream, (const ulonglong2*)in, (ulonglong2*)out, flags, rows, n_instances, (u32)mont, fc_dev); break;
            case 3: hipLaunchKernelGGL(h2e_export_columns<3>, gridc, block, 0, stream, (const ulonglong2*)in, (ulonglong2*)out, flags, rows, n_instances, (u32)mont, fc_dev); break;
            case 2: hipLaunchKernelGGL(h2e_export_columns<2>, gridc, block, 0, stream, (const ulonglong2*)in, (ulonglong2*)out, flags, rows, n_instances, (u32)mont, fc_dev); break;
            default: return -1;
        }
        return (int)hipGetLastError();
    }
#define H2E_EXPORT(C, COLMAJ)                                                                                                  \
    hipLaunchKernelGGL((h2e_export<C, COLMAJ>), grid, block, 0, stream, (const ulonglong2*)in, (ulonglong2*)out, flags, rows, \
                       n_instances, (u32)mont, fc_dev)
    switch (cols) {
        case 5: if (columns) H2E_EXPORT(5, true); else H2E_EXPORT(5, false); break;
        case 3: if (columns) H2E_EXPORT(3, true); else H2E_EXPORT(3, false); break;
        case 2: if (columns) H2E_EXPORT(2, true); else H2E_EXPORT(2, false); break;
        default: return -1;
    }
#undef H2E_EXPORT
    return (int)hipGetLastError();
}

#endif   // H2E_COMMON_UNIT
// ------------------------------------------------------------------------------------------------
// host-callable launcher (C linkage, used by the C-ABI layer in h2e_capi.cpp)
extern "C" int H2E_UNIT(h2e_engine_set_consts)(int field_pair, const H2EFieldConsts* host) {
    if (field_pair < 0 || field_pair > 2) return -1;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fc), host, sizeof(H2EFieldConsts), (size_t)field_pair * sizeof(H2EFieldConsts),
                                  hipMemcpyHostToDevice);
}

#if H2E_COMMON_UNIT
// ------------------------------------------------------------------------------------------------
// On-device consumer of a streaming job (SURVEY.md 8d cfg 3 / 8e): a 32-byte digest per instance of one region's
// batch-interleaved advice array, so that tiles can be checked / gathered without their 1.2 GB leaving the GPU.
//   digest[j] = sum over assigned cells (row, col) of  sm(w_j ^ sm(row * COLS + col) ^ j * 0xA24BAED4963EE407)   (mod 2^64)
// with sm = the SplitMix64 finaliser and w_0..w_3 the cell's words: a sum, so the order cells are visited in does not
// matter (oracle/digest.hpp computes the same over its Records).  HBM bound: every cell is read once; a wave reads one
// (row, col, half) of 64 instances per load (1 KB contiguous).
WI_INLINE u64 h2e_sm64(u64 z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <int COLS>
__global__ void __launch_bounds__(256) h2e_digest(const ulonglong2* __restrict__ in, const uint8_t* __restrict__ flags, u64 rows,
                                                  u32 n_inst, u64* __restrict__ out) {
    constexpr u32 TR = 64;   // rows per tile; the 4 waves of a block take every 4th row of a tile
    u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    u64 tiles = (rows + TR - 1) / TR;
    for (u32 i0 = 0; i0 < n_inst; i0 += 64) {
        u32 inst = i0 + lane;
        bool live = inst < n_inst;
        u64 d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        for (u64 t = blockIdx.x; t < tiles; t += gridDim.x) {
            u64 r1 = min(rows, (t + 1) * TR);
            for (u64 row = t * TR + wave; row < r1; row += 4) {
                // all of the row's loads first (up to 2 x COLS independent 1 KB wave loads in flight), then the hashing
                ulonglong2 lo[COLS], hi[COLS];
                bool on[COLS];
#pragma unroll
                for (int col = 0; col < COLS; col++) {
                    u64 cell = row * COLS + col;
                    on[col] = live && (flags == nullptr || (flags[cell] & 1));   // the flag is wave-uniform
                    if (on[col]) {
                        lo[col] = in[(cell * 2) * n_inst + inst];
                        hi[col] = in[(cell * 2 + 1) * n_inst + inst];
                    }
                }
#pragma unroll
                for (int col = 0; col < COLS; col++) {
                    if (!on[col]) continue;
                    u64 tt = h2e_sm64(row * COLS + col);
                    d0 += h2e_sm64(lo[col].x ^ tt);
                    d1 += h2e_sm64(lo[col].y ^ tt ^ 0xA24BAED4963EE407ull);
                    d2 += h2e_sm64(hi[col].x ^ tt ^ (2 * 0xA24BAED4963EE407ull));
                    d3 += h2e_sm64(hi[col].y ^ tt ^ (3 * 0xA24BAED4963EE407ull));
                }
            }
        }
        if (live) {
            atomicAdd((unsigned long long*)&out[(size_t)inst * 4 + 0], (unsigned long long)d0);
            atomicAdd((unsigned long long*)&out[(size_t)inst * 4 + 1], (unsigned long long)d1);
            atomicAdd((unsigned long long*)&out[(size_t)inst * 4 + 2], (unsigned long long)d2);
            atomicAdd((unsigned long long*)&out[(size_t)inst * 4 + 3], (unsigned long long)d3);
        }
    }
}
extern "C" int h2e_engine_digest(uint32_t cols, const void* in, const uint8_t* flags, uint64_t rows, uint32_t n_instances, void* out,
                                 hipStream_t stream) {
    if (n_instances == 0) return 0;
    hipError_t e = hipMemsetAsync(out, 0, (size_t)n_instances * 32, stream);
    if (e != hipSuccess) return (int)e;
    if (rows == 0) return 0;
    u64 tiles = (rows + 63) / 64;
    dim3 grid((u32)std::min<u64>(tiles, 2048)), block(256);
    switch (cols) {
        case 5: hipLaunchKernelGGL(h2e_digest<5>, grid, block, 0, stream, (const ulonglong2*)in, flags, rows, n_instances, (u64*)out); break;
        case 3: hipLaunchKernelGGL(h2e_digest<3>, grid, block, 0, stream, (const ulonglong2*)in, flags, rows, n_instances, (u64*)out); break;
        case 2: hipLaunchKernelGGL(h2e_digest<2>, grid, block, 0, stream, (const ulonglong2*)in, flags, rows, n_instances, (u64*)out); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Shape artefacts in the prover's layout, generated on the device.
// (a) fixed columns: the program's fixed cells are dictionary ids (h2e_shape); expanded to [instance][COLS][row][4 words]
//     (column-major, what halo2's fixed columns are) or [instance][row][COLS][4], None -> 0, canonical or Montgomery form.
template <bool COLUMNS>
__global__ void __launch_bounds__(256) h2e_fixed_columns(const u32* __restrict__ ids, const u64* __restrict__ dict, u64 rows, u32 cols,
                                                         u32 n_inst, u32 mont, const H2EFieldConsts* fc, ulonglong2* __restrict__ out) {
    u64 cell = (u64)blockIdx.x * 256 + threadIdx.x;
    if (cell >= rows * cols) return;
    u64 row = cell / cols;
    u32 col = (u32)(cell % cols);
    u32 id = ids[cell];
    Fe v = wd_load<4>(dict + (size_t)id * 4);   // entry 0 = None = zero
    if (mont) {
        Mont<4> M = mont_n(fc);
        v = mont_mul<4>(M, v, M.r2);
    }
    u64 o = COLUMNS ? (u64)col * rows + row : cell;
    for (u32 i = 0; i < n_inst; i++) {
        ulonglong2* q = out + ((u64)i * rows * cols + o) * 2;
        q[0] = make_ulonglong2(v.v[0], v.v[1]);
        q[1] = make_ulonglong2(v.v[2], v.v[3]);
    }
}
// constants made from instance inputs (the G2 points of a pairing check): [row, fixed col, input slot, limb (-1: value mod n)]
template <class FP>
__global__ void h2e_fixed_patches(const u32* __restrict__ patches, u32 n_patches, const u64* __restrict__ inputs, u32 n_slots, u32 slot_words,
                                  u64 rows, u32 cols, u32 columns, u32 n_inst, u32 mont, const H2EFieldConsts* fc, ulonglong2* __restrict__ out) {
    u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_patches * n_inst) return;
    u32 k = gid % n_patches, i = gid / n_patches;
    u32 row = patches[4 * k], col = patches[4 * k + 1], slot = patches[4 * k + 2];
    int limb = (int)patches[4 * k + 3];
    Wd<FP::WW> x = wd_load<FP::WW>(inputs + ((size_t)i * n_slots + slot) * slot_words);
    LC c;
    c.fc = fc;
    Fe v;
    if (limb < 0) {
        v = mod_n<FP::WW>(c, x);
    } else {
        Limb l[FP::L];
        split_limbs<FP>(x, l);
        Limb pick = l[0];
#pragma unroll
        for (int j = 1; j < FP::L; j++)
            if (j == limb) pick = l[j];
        v = fe_of(pick);
    }
    if (mont) {
        Mont<4> M = mont_n(fc);
        v = mont_mul<4>(M, v, M.r2);
    }
    // rows = 0: the values alone, [instance][patch][4 words] (h2e_engine_patch_values: the device-side constraint check)
    u64 o = rows == 0 ? (u64)k : columns ? (u64)col * rows + row : (u64)row * cols + col;
    ulonglong2* q = out + ((u64)i * (rows == 0 ? (u64)n_patches : rows * cols) + o) * 2;
    q[0] = make_ulonglong2(v.v[0], v.v[1]);
    q[1] = make_ulonglong2(v.v[2], v.v[3]);
}
// (b) the 18-bit tagged range lookup table (RangeChip::init_table, src/circuit/range_chip.rs:230-258): for tag in 0..=18,
//     value in 0..2^tag: row (tag, value); 2^19 - 1 rows, two columns, column-major [2][rows][4 words]
__global__ void h2e_range_table(u32 mont, const H2EFieldConsts* fc, ulonglong2* __restrict__ out) {
    const u32 rows = (1u << 19) - 1;
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    u32 tag = 31 - __clz(r + 1), value = r + 1 - (1u << tag);
    Fe t = fe_u64(tag), v = fe_u64(value);
    if (mont) {
        Mont<4> M = mont_n(fc);
        t = mont_mul<4>(M, t, M.r2);
        v = mont_mul<4>(M, v, M.r2);
    }
    out[(size_t)r * 2] = make_ulonglong2(t.v[0], t.v[1]);
    out[(size_t)r * 2 + 1] = make_ulonglong2(t.v[2], t.v[3]);
    out[((size_t)rows + r) * 2] = make_ulonglong2(v.v[0], v.v[1]);
    out[((size_t)rows + r) * 2 + 1] = make_ulonglong2(v.v[2], v.v[3]);
}
// (c) the permutation list as copy constraints between (advice column, row) pairs: global advice column = 0-4 base, 5-7 range,
//     8-9 select; out[k] = [column_a, row_a, column_b, row_b]
__global__ void h2e_copy_constraints(const u32* __restrict__ perms, u64 n, uint4* __restrict__ out) {
    u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    u32 a = perms[2 * k], b = perms[2 * k + 1];
    const u32 base_col[3] = {0, 5, 8};
    out[k] = make_uint4(base_col[H2E_REF_REGION(a)] + H2E_REF_COL(a), H2E_REF_ROW(a), base_col[H2E_REF_REGION(b)] + H2E_REF_COL(b), H2E_REF_ROW(b));
}
extern "C" int h2e_engine_fixed(int field_pair, const uint32_t* ids, const uint64_t* dict, uint64_t rows, uint32_t cols, int columns, int mont,
                                const uint32_t* patches, uint32_t n_patches, const uint64_t* inputs, uint32_t n_slots, uint32_t slot_words,
                                uint32_t n_instances, const H2EFieldConsts* fc_dev, void* out, hipStream_t stream) {
    if (rows == 0 || n_instances == 0) return 0;
    u64 cells = rows * cols;
    if ((cells + 255) / 256 > 0x7fffffffull) return -1;
    dim3 grid((u32)((cells + 255) / 256)), block(256);
    if (columns) hipLaunchKernelGGL((h2e_fixed_columns<true>), grid, block, 0, stream, ids, dict, rows, cols, n_instances, (u32)mont, fc_dev, (ulonglong2*)out);
    else hipLaunchKernelGGL((h2e_fixed_columns<false>), grid, block, 0, stream, ids, dict, rows, cols, n_instances, (u32)mont, fc_dev, (ulonglong2*)out);
    if (n_patches && inputs) {
        dim3 g2((n_patches * n_instances + 63) / 64), b2(64);
#define H2E_PATCH(FP)                                                                                                               \
    hipLaunchKernelGGL(h2e_fixed_patches<FP>, g2, b2, 0, stream, patches, n_patches, inputs, n_slots, slot_words, rows, cols, (u32)columns, \
                       n_instances, (u32)mont, fc_dev, (ulonglong2*)out)
        switch (field_pair) {
            case 0: H2E_PATCH(FP_BN256_FQ); break;
            case 1: H2E_PATCH(FP_BLS_FQ); break;
            case 2: H2E_PATCH(FP_BLS_FR); break;
            default: return -1;
        }
#undef H2E_PATCH
    }
    return (int)hipGetLastError();
}
extern "C" int h2e_engine_patch_values(int field_pair, const uint32_t* patches, uint32_t n_patches, const uint64_t* inputs, uint32_t n_slots,
                                       uint32_t slot_words, uint32_t n_instances, const H2EFieldConsts* fc_dev, void* out, hipStream_t stream) {
    if (n_patches == 0 || n_instances == 0) return 0;
    dim3 g2((n_patches * n_instances + 63) / 64), b2(64);
#define H2E_PATCHV(FP)                                                                                                                  \
    hipLaunchKernelGGL(h2e_fixed_patches<FP>, g2, b2, 0, stream, patches, n_patches, inputs, n_slots, slot_words, (u64)0, 0u, 0u, n_instances, 0u, \
                       fc_dev, (ulonglong2*)out)
    switch (field_pair) {
        case 0: H2E_PATCHV(FP_BN256_FQ); break;
        case 1: H2E_PATCHV(FP_BLS_FQ); break;
        case 2: H2E_PATCHV(FP_BLS_FR); break;
        default: return -1;
    }
#undef H2E_PATCHV
    return (int)hipGetLastError();
}
extern "C" int h2e_engine_range_table(int mont, const H2EFieldConsts* fc_dev, void* out, hipStream_t stream) {
    hipLaunchKernelGGL(h2e_range_table, dim3(((1u << 19) - 1 + 255) / 256), dim3(256), 0, stream, (u32)mont, fc_dev, (ulonglong2*)out);
    return (int)hipGetLastError();
}
extern "C" int h2e_engine_copy_constraints(const uint32_t* perms, uint64_t n, void* out, hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(h2e_copy_constraints, dim3((u32)((n + 255) / 256)), dim3(256), 0, stream, perms, n, (uint4*)out);
    return (int)hipGetLastError();
}

__global__ void h2e_or_status(const InstanceDesc* inst, u32 n_instances, u32 bits) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_instances) atomicOr(inst[i].status, bits);
}
// the run's stream digest: the sum of its shards (tape.h H2ELaunch::dg_out)
__global__ void __launch_bounds__(256) h2e_digest_reduce(const u64* __restrict__ shards, u32 n_shards, u32 n_words, u64* __restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    u64 s = 0;
    for (u32 k = 0; k < n_shards; k++) s += shards[(size_t)k * n_words + i];
    out[i] = s;
}
extern "C" int h2e_engine_digest_reduce(const void* shards, uint32_t n_shards, uint32_t n_words, void* out, hipStream_t stream) {
    hipLaunchKernelGGL(h2e_digest_reduce, dim3((n_words + 255) / 256), dim3(256), 0, stream, (const u64*)shards, n_shards, n_words, (u64*)out);
    return (int)hipGetLastError();
}
extern "C" int h2e_engine_or_status(const void* instances, uint32_t n_instances, uint32_t bits, hipStream_t stream) {
    if (n_instances == 0) return 0;
    hipLaunchKernelGGL(h2e_or_status, dim3((n_instances + 63) / 64), dim3(64), 0, stream, (const InstanceDesc*)instances, n_instances, bits);
    return (int)hipGetLastError();
}

#endif   // H2E_COMMON_UNIT
// ------------------------------------------------------------------------------------------------
// Phase 1 of the tail's scan in digit rows (round 5).  The doubling chain over the 254 windows is inherently serial - window 0's
// sum is doubled 253 times - and as one lane per instance (h2e_predict_tail phase 1: ONE wave for 64 instances, 2 034 Montgomery
// products of ~330 instructions each) it was 3.5-4 ms of every run's tail, on the path to the run's completion.  Here an instance is
// a 16-lane row, a lane a 32-bit digit (DigitRow: the pairings' arithmetic): a product is ~90 wave instructions for four instances,
// additions and subtractions are column sums with one lazy reduction ([0, 2 w) throughout; what is stored for the lane kernels is
// brought to [0, w)), 16 waves on 16 CUs instead of one.  Same formulas as jac_dbl / jac_add, so the values stored are the lane
// kernel's.  An instance that meets a degenerate sum (Z = 0; or the test knob) only raises its flag: h2e_predict_tail phase 3 then
// redoes that instance the old way, fallback walk included.
template <class FP>
__global__ void __launch_bounds__(64) h2e_predict_tail_rows(H2EPreKernel K, const u32* __restrict__ args, const InstanceDesc* __restrict__ inst, u32 n_instances) {
    constexpr int N = FP::WW, D = 2 * N;
    constexpr u32 CH = H2E_TAIL_CHUNK;
    __builtin_amdgcn_s_setprio(3);
    const u32 lane = threadIdx.x & 63u;
    u32 instance = blockIdx.x * 4u + (lane >> 4);
    const bool live = instance < n_instances;
    if (!live) instance = n_instances - 1u;
    InstanceDesc d = inst[instance];
    const H2EFieldConsts* fc = &g_fc[FP::ID];
    DigitRow<D> R = DigitRow<D>::make(fc, lane);
    const u32 j = R.j;
    const bool digit_lane = j < (u32)D;
    const u32 jd = digit_lane ? j : 0u;
    const u32 r1j = digit_lane ? ((const H2E_AS_GLOBAL u32*)fc->w_r1)[jd] : 0u;
    const i64 beta = digit_lane ? (i64)((const H2E_AS_GLOBAL u64*)fc->lin_bias)[jd] : 0;
    const u32 minv32 = (u32)fc->w_minv;
    const u32* a = args + K.args_begin;
    const u32 windows = a[0];
    const u32 nch = (windows + CH - 1) / CH;
    const u32 sB = K.scan_begin, sD = sB + windows, sA = sD + windows, sX = sA + nch;
    auto cptr = [&](u32 slot, u32 coord) { return (H2E_AS_GLOBAL u32*)(d.jac + ((size_t)slot * 3 + coord) * d.ws); };
    auto ldc = [&](u32 slot, u32 coord) -> u32 { return digit_lane ? cptr(slot, coord)[j] : 0u; };
    auto mul = [&](u32 x, u32 y) -> u32 { return R.mont_mul(x, y, minv32); };
    // c1 x1 + c2 x2 + c3 x3 mod w in [0, 2 w) (|c| <= 8: the columns' bias covers far more)
    auto lin = [&](int c1, u32 x1, int c2, u32 x2, int c3, u32 x3) -> u32 {
        i64 acc = beta + (i64)c1 * (i64)(u64)x1 + (i64)c2 * (i64)(u64)x2 + (i64)c3 * (i64)(u64)x3;
        return R.reduce_columns((u32)(u64)acc, (u32)((u64)acc >> 32));
    };
    // [0, 2 w) -> [0, w): x - w + 2^(32 D) digit by digit; its carry out of the top digit (lane D) says x >= w
    auto canon = [&](u32 x) -> u32 {
        u64 s2 = (u64)x + (u64)(digit_lane ? ~R.wj : 0u) + (j == 0u ? 1ull : 0ull);
        u64 G = __builtin_amdgcn_ballot_w64((u32)(s2 >> 32) != 0u);
        u32 t = DigitRow<D>::carry((u32)s2, G);
        u32 ge = (u32)__builtin_amdgcn_update_dpp(0, (int)t, H2E_DPP_ROW_BCAST(D), 0xf, 0xf, true);
        return digit_lane ? (ge ? t : x) : 0u;
    };
    auto is_zero = [&](u32 x) -> bool {   // a value in [0, 2 w): zero is 0 or w (row-uniform result)
        u64 nz = __builtin_amdgcn_ballot_w64(x != 0u), nw = __builtin_amdgcn_ballot_w64(x != R.wj);
        const u32 row_base = lane & 48u;
        return ((u32)(nz >> row_base) & 0xffffu) == 0u || ((u32)(nw >> row_base) & 0xffffu) == 0u;
    };
    struct P3 {
        u32 x, y, z;
    };
    auto dbl = [&](const P3& p) -> P3 {   // jac_dbl
        u32 aa = mul(p.x, p.x), b = mul(p.y, p.y), cc = mul(b, b);
        u32 xb = lin(1, p.x, 1, b, 0, 0u);
        u32 dq = lin(2, mul(xb, xb), -2, aa, -2, cc);
        u32 e = lin(3, aa, 0, 0u, 0, 0u);
        P3 r;
        r.x = lin(1, mul(e, e), -2, dq, 0, 0u);
        r.y = lin(1, mul(e, lin(1, dq, -1, r.x, 0, 0u)), -8, cc, 0, 0u);
        r.z = lin(2, mul(p.y, p.z), 0, 0u, 0, 0u);
        return r;
    };
    auto add = [&](const P3& p, const P3& q) -> P3 {   // jac_add
        u32 z1z1 = mul(p.z, p.z), z2z2 = mul(q.z, q.z);
        u32 u1 = mul(p.x, z2z2), u2 = mul(q.x, z1z1);
        u32 s1 = mul(mul(p.y, q.z), z2z2), s2 = mul(mul(q.y, p.z), z1z1);
        u32 h = lin(1, u2, -1, u1, 0, 0u), r = lin(1, s2, -1, s1, 0, 0u);
        u32 hh = mul(h, h), hhh = mul(hh, h), vv = mul(u1, hh);
        P3 o;
        o.x = lin(1, mul(r, r), -1, hhh, -2, vv);
        o.y = lin(1, mul(r, lin(1, vv, -1, o.x, 0, 0u)), -1, mul(s1, hhh), 0, 0u);
        o.z = mul(mul(p.z, q.z), h);
        return o;
    };
    auto store = [&](u32 slot, const P3& p) {
        u32 cx = canon(p.x), cy = canon(p.y), cz = canon(p.z);
        if (live && digit_lane) {
            cptr(slot, 0)[j] = cx;
            cptr(slot, 1)[j] = cy;
            cptr(slot, 2)[j] = cz;
        }
    };
    P3 acc;
    acc.x = ldc(sX, 0);
    acc.y = ldc(sX, 1);
    acc.z = r1j;
    bool fallback = false;
    for (u32 c = 0; c < nch; c++) {
        const u32 w0 = c * CH, w1 = min(w0 + CH, windows);
        P3 dd = acc;
        for (u32 w = w0; w < w1; w++) {
            dd = dbl(dd);
            store(sD + w, dd);
        }
        P3 b;
        b.x = ldc(sB + w1 - 1, 0);
        b.y = ldc(sB + w1 - 1, 1);
        b.z = ldc(sB + w1 - 1, 2);
        P3 nx = add(b, dd);
        if (is_zero(nx.z) || ((g_scan_test & 2u) && (c & 1u))) fallback = true;
        acc = nx;
        store(sA + c, acc);
    }
    if (fallback && live && j == 0u) cptr(sX, 2)[0] = 1u;
}

// Tuning knobs (h2e_capi.cpp reads H2E_TUNE once, at h2e_ctx_create): [0] LDS bytes a small predictor grid reserves so
// that no expansion wave shares its CU, [1] expansion result cache in LDS on / off, [2] extra dynamic LDS per expansion
// workgroup (caps its waves per CU).  Round 2 (profiles/r2_tune_sweep.txt) switched the result cache off:
// with the batch-interleaved layout an operand re-read is a coalesced 1 KB load and the cache bought the expansion
// nothing (11.6 vs 11.7 ms), while the 15 KB of LDS per workgroup it held kept the value chain's replay workgroups
// (95-135 KB of LDS each) of the next run from sharing CUs with the expansion: pipelined step 24.0 -> 20.9 ms.
// [1] bit 1: the expansion's waves run at the chain kernels' priority (s_setprio 3) - the shared expansion stream is the
// pipelined step's busiest resource: 16.17 -> 16.02 ms, window expansion 11.85 -> 11.4 ms; on by default.
// [4]: persistent expansion (experiment): big expansions are launched with this many workgroups per CU, each looping over its share
// of the blocks (0 = one workgroup per block)
// [5]: 1 = no packed expansion for batches smaller than a wave (h2e_run_tape_packed; A/B), 2 = packed in tape order (no order tables)
// [1] bit 0, the result cache, is ON again since round 4: the windows' and the accumulation loop's replays - 56-135 KB of LDS per
// workgroup, the neighbours its 15 KB per wave kept off the expansion's CUs - are hint stores now (no LDS).  64 x 1024-point tiles,
// alternating with it off in one box: step 15.38-15.42 -> 15.12-15.24 ms, window expansion 0.70 -> 0.71, traffic of a window launch
// 36.95 -> 35.15 GB = 1.08 x its algorithmic bytes (the re-reads of operands the previous ops just produced).
static int g_tune[6] = {0, 3, 0, 0, 0, 0};
extern "C" long long H2E_UNIT(h2e_engine_scan_fallbacks)(void) {
    unsigned long long n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_scan_fallbacks), sizeof(n)) != hipSuccess) return -1;
    return (long long)n;
}
extern "C" void H2E_UNIT(h2e_engine_set_tuning)(int key, int value) {
    if (key >= 0 && key < 6) g_tune[key] = value;   // ([3]: also the device-side scan test mask below; bit 4 = the tail's phase 1 as one lane per instance, A/B)
    if (key == 3) {
        u32 m = (u32)value;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_scan_test), &m, sizeof(m));
    }
}

// dynamic LDS of h2e_run_tape_packed in front of its result cache: op buffer (G groups x min(32, 256 / G) ops x 64 bytes) + digest sums
static size_t pk_lds_bytes(int log2p, bool digest) {
    const u32 G = 64u >> log2p, CH = std::min(32u, H2E_PK_BUF_OPS / G);
    return (size_t)G * CH * 64u + (digest ? (size_t)12 * 64 * 8 : 0);
}
// mode: 1 = values-only replay (whole tape per lane), 2 = full expansion (sub-ranges if any), 4 = inverse fix-up
extern "C" int H2E_UNIT(h2e_engine_launch)(int field_pair, int mode, const H2ELaunch* launch, const void* instances,
                                 uint32_t n_instances, const H2EFieldConsts* fc_dev, hipStream_t stream) {
    u32 per_sub = n_instances * launch->n_strands;
    if (per_sub == 0 || launch->n_ops == 0) return 0;
    u32 blocks_per_sub = (per_sub + 63) / 64;
    u32 n_sub = launch->n_sub > 1 ? launch->n_sub : 1;
    dim3 block(64), grid1(blocks_per_sub), grid(blocks_per_sub * n_sub);
    const InstanceDesc* inst = (const InstanceDesc*)instances;
    // batches smaller than half a wave: several sub-ranges per wave (h2e_run_tape_packed); g_tune[5] = 1 switches it off (A/B)
    int pack_log2p = -1;
    if (per_sub <= 32 && n_sub >= 2 && g_tune[5] != 1) {
        pack_log2p = 1;   // (at most 32 groups per wave: the kernel's op buffer holds that many chunks)
        while ((1u << pack_log2p) < per_sub) pack_log2p++;
    }
    // the result cache: on request - and for the packed form, whose launches are too small for its LDS to be in anybody's way
    // (16 x bls12_381: 1.32 -> 1.30 ms, 2 x: 0.58 -> 0.52, 8 x bn256: 0.69 -> 0.61)
    const bool xcache_on = (g_tune[1] & 1) != 0 || pack_log2p >= 0;
    H2ELaunch launch_x = *launch;
    // the packed form's order table for this group count (tape.h pk_order; g_tune[5] = 2: tape order, A/B)
    u32 pk_grid = (n_sub + (64u >> (pack_log2p < 0 ? 0 : pack_log2p)) - 1) / (64u >> (pack_log2p < 0 ? 0 : pack_log2p));
    launch_x.pk_order = nullptr;
    if (pack_log2p >= 1 && pack_log2p <= 5 && launch->pk_order && launch->pk_n_sub == launch->n_sub && launch->pk_waves[5 - pack_log2p] &&
        g_tune[5] != 2) {
        launch_x.pk_order = launch->pk_order + launch->pk_off[5 - pack_log2p];
        pk_grid = launch->pk_waves[5 - pack_log2p];
    }
    if (xcache_on) launch_x.rel_refs |= 4u;
    if (g_tune[1] & 2) launch_x.rel_refs |= 8u;
    dim3 grid_x = grid;
    launch_x.x_blocks = 0;
    if (g_tune[4] > 0 && grid.x > 16384) {
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
            if (n_cu <= 0) n_cu = 256;
        }
        launch_x.x_blocks = grid.x;
        grid_x = dim3((u32)n_cu * (u32)g_tune[4]);
    }
#ifdef H2E_AB_KERNELS
#define H2E_AB_REPLAY_WAVE(FP)                                                                                                 \
    if ((mode & 1) && launch->lrecs && launch->l_pair == 2) {                                                                   \
        hipLaunchKernelGGL(h2e_replay_wave<FP>, dim3(n_instances * launch->n_strands), dim3(64),                                \
                           (size_t)2 * H2E_WCHUNK * 32 + (64 * H2E_MAX_L * 2 + 64 * 4) * 8 + (size_t)launch->l_slots * LVals<FP>::W * 8, \
                           stream, *launch, inst, n_instances);                                                                \
        mode &= ~1;                                                                                                            \
    }
#else
#define H2E_AB_REPLAY_WAVE(FP) \
    if ((mode & 1) && launch->lrecs && launch->l_pair == 2) return -4;   /* a program form only A/B builds have a kernel for */
#endif
#define H2E_LAUNCH_FP(FP)                                                                                                     \
    if ((mode & 1) && launch->s_words) {                                                                                        \
        u32 lanes = launch->n_sops * launch->n_strands * n_instances;                                                          \
        if (lanes) hipLaunchKernelGGL(h2e_hint_store<FP>, dim3((lanes + 63) / 64), dim3(64), 0, stream, *launch, inst, n_instances); \
        mode &= ~1;                                                                                                            \
    }                                                                                                                          \
    H2E_AB_REPLAY_WAVE(FP)                                                                                                     \
    if ((mode & 1) && launch->lrecs) {                                                                                         \
        hipLaunchKernelGGL(h2e_replay_levels<FP>, dim3((n_instances * launch->n_strands + (launch->l_pair ? 1 : 0)) / (launch->l_pair ? 2 : 1)), \
                           dim3(64 * H2E_LEVEL_WAVES), (size_t)launch->l_slots * LVals<FP>::W * 8 * (launch->l_pair ? 2 : 1), stream, \
                           *launch, inst, n_instances);                                                                       \
        mode &= ~1;                                                                                                            \
    }                                                                                                                          \
    if ((mode & 1) && launch->vtape &&                                                                                         \
        ((size_t)launch->v_units * 2 + ((size_t)launch->v_int_slots * VSlots<FP>::W + VSlots<FP>::NF * 4)) * 64 * 8 +           \
                2 * H2E_VCHUNK * sizeof(H2EVRec) > 160 * 1024)                                                  \
        return -3;   /* the host compiler's LDS budget and the kernel's static LDS disagree */                                 \
    if ((mode & 1) && launch->vtape)                                                                                           \
        hipLaunchKernelGGL(h2e_replay<FP>, dim3(blocks_per_sub * launch->n_vpieces), block,                                    \
                           ((size_t)launch->v_units * 2 + ((size_t)launch->v_int_slots * VSlots<FP>::W + VSlots<FP>::NF * 4)) * 64 * 8, \
                           stream, *launch, inst, n_instances);                                                                \
    if ((mode & 1) && !launch->vtape) return -2;   /* a values-only replay always runs from the compiled V-tape */          \
    if ((mode & 2) && pack_log2p >= 0)                                                                                         \
        hipLaunchKernelGGL(h2e_run_tape_packed<FP>, dim3(pk_grid), block,                                                      \
                           pk_lds_bytes(pack_log2p, launch->dg_out != nullptr) + (xcache_on ? (size_t)3 * (2 * FP::L + 4) * 64 * 8 : 0), \
                           stream, launch_x, inst, n_instances, (u32)pack_log2p);                                              \
    else if (mode & 2)                                                                                                         \
        hipLaunchKernelGGL((h2e_run_tape<FP, false>), grid_x, block,                                                           \
                           (xcache_on ? (size_t)3 * (2 * FP::L + 4) * 64 * 8 : 0) + (launch->dg_out ? (size_t)12 * 64 * 8 : 0) +  \
                               (grid.x > 4096 ? (size_t)g_tune[2] : 0),                                                          \
                           stream, launch_x, inst, n_instances, fc_dev);
    switch (field_pair) {
#if H2E_HAS_FP(0)
        case 0: { H2E_LAUNCH_FP(FP_BN256_FQ) } break;
#endif
#if H2E_HAS_FP(1)
        case 1: { H2E_LAUNCH_FP(FP_BLS_FQ) } break;
#endif
#if H2E_HAS_FP(2)
        case 2: { H2E_LAUNCH_FP(FP_BLS_FR) } break;
#endif
        default: return -1;
    }
#undef H2E_LAUNCH_FP
    {
        hipError_t le = hipGetLastError();   // (reading it resets it: read once)
        if (le != hipSuccess) return (int)le;
    }
    if ((mode & 4) && launch->n_fixups) {
        u32 chunks = (launch->n_fixups + FIXUP_K - 1) / FIXUP_K;
        u32 lanes = n_instances * launch->n_strands * chunks;
        hipLaunchKernelGGL(h2e_fixup_inverses, dim3((lanes + 63) / 64), dim3(64), 0, stream, *launch, inst, n_instances, fc_dev);
    }
    return (int)hipGetLastError();
}

// phase: 1 = the predictor chain, 2 = its finalize (batch inversion -> hints), 3 = both
extern "C" int H2E_UNIT(h2e_engine_predict)(int field_pair, int phase, const H2EPreKernel* k, const uint32_t* args_dev, const uint32_t* params_dev,
                                  const uint32_t* aux_dev, const void* instances, uint32_t n_instances,
                                  const H2EFieldConsts* fc_dev, hipStream_t stream) {
    const InstanceDesc* inst = (const InstanceDesc*)instances;
    u32 lanes = n_instances * k->n_lanes;
    if (lanes == 0) return 0;
    dim3 block(64), grid((lanes + 63) / 64);
    // A latency-bound predictor (few waves) must not share its CU with expansion waves of another stream: ask
    // for most of the CU's LDS so that nothing else fits next to it.
    size_t lds_reserve = grid.x <= 512 ? (size_t)g_tune[0] : 0;
    u32 n_hints = k->n_lanes * k->hints_per_lane;
    u32 chunks = (n_hints + HINT_K - 1) / HINT_K;
    dim3 grid2((n_instances * chunks + 63) / 64);
    u32 ecc_chunks = (k->ecc_ops + ECC_CH - 1) / ECC_CH;
    dim3 grid3((n_instances * k->n_lanes * ecc_chunks + 63) / 64);
#ifdef H2E_AB_KERNELS
#define H2E_AB_FIELD_LANES(FP)                                                                                                     \
    hipLaunchKernelGGL(h2e_field_chain<FP>, dim3(n_instances), dim3(128), lds, stream, *k, args_dev, (const u64*)params_dev, inst, n_instances);
#else
#define H2E_AB_FIELD_LANES(FP) return -4;   /* the lane-per-record form: a kernel of A/B builds only */
#endif
#define H2E_PREDICT_FP(FP)                                                                                                          \
    if (k->kind == H2E_PRE_FIELD_CHAIN) {   /* params_dev carries the constant pool, n_params the words per input slot */          \
        if (phase & 1)                                                                                                              \
        {                                                                                                                           \
            size_t lds = (k->f_mode == 1 ? (size_t)H2E_DP_CHUNKS * H2E_WCHUNK * 64 + H2E_DP_DIV_SCRATCH : (size_t)2 * H2E_WCHUNK * 32) + (size_t)k->f_slots * FP::WW * 8 + 64; \
            if (k->f_mode == 1 && (size_t)g_tune[0] > lds && g_tune[0] <= 160 * 1024) lds = (size_t)g_tune[0];   /* keep the CU (A/B) */ \
            if (k->f_mode == 1)                                                                                                     \
                hipLaunchKernelGGL(h2e_field_chain_digits<FP>, dim3(n_instances), dim3((H2E_DP_WAVES + 1) * 64), lds, stream, *k, args_dev, \
                                   (const u64*)params_dev, inst, n_instances);                                                     \
            else                                                                                                                    \
                H2E_AB_FIELD_LANES(FP)                                                                                              \
        }                                                                                                                           \
        if ((phase & 2) && k->hints_per_lane)                                                                                       \
            hipLaunchKernelGGL(h2e_field_finalize<FP>, dim3((n_instances * k->hints_per_lane + 63) / 64), block, 0, stream,         \
                               k->hint_base, k->hints_per_lane, inst, n_instances);                                                \
        if ((phase & 2) && k->hints2_per_lane)                                                                                      \
            hipLaunchKernelGGL(h2e_field_finalize<FP>, dim3((n_instances * k->hints2_per_lane + 63) / 64), block, 0, stream,        \
                               k->hint2_base, k->hints2_per_lane, inst, n_instances);                                              \
        if ((phase & 2) && k->f_n_sinks)                                                                                            \
            hipLaunchKernelGGL(h2e_field_sinks<FP>, dim3((n_instances * k->f_n_sinks + 63) / 64), block, 0, stream, *k, args_dev,    \
                               (const u64*)params_dev, inst, n_instances);                                                         \
        break;                                                                                                                      \
    }                                                                                                                               \
    if (k->kind == H2E_PRE_MSM_SELECT) {                                                                                            \
        if (phase & 1) hipLaunchKernelGGL(h2e_select<FP>, grid, block, 0, stream, *k, args_dev, params_dev, aux_dev, inst, n_instances, fc_dev); \
        break;                                                                                                                      \
    }                                                                                                                               \
    if ((phase & 1) && k->kind == H2E_PRE_MSM_WINDOWS) {                                                                            \
        u32 ng = k->ecc_ops, len = (ng + H2E_WIN_CHUNKS - 1) / H2E_WIN_CHUNKS, nch = (ng + len - 1) / len;                          \
        dim3 gc((lanes * nch + 63) / 64);                                                                                           \
        hipLaunchKernelGGL(h2e_predict_windows<FP>, gc, block, 0, stream, *k, 0u, args_dev, inst, n_instances, fc_dev);             \
        hipLaunchKernelGGL(h2e_predict_windows<FP>, grid, block, 0, stream, *k, 1u, args_dev, inst, n_instances, fc_dev);           \
        hipLaunchKernelGGL(h2e_predict_windows<FP>, gc, block, 0, stream, *k, 2u, args_dev, inst, n_instances, fc_dev);             \
    } else if ((phase & 1) && k->kind == H2E_PRE_MSM_TAIL) {                                                                        \
        u32 windows = k->ecc_ops / k->pattern_len, nch = (windows + H2E_TAIL_CHUNK - 1) / H2E_TAIL_CHUNK;                           \
        dim3 g0((n_instances * nch + 63) / 64), g2((n_instances * windows + 63) / 64);                                              \
        hipLaunchKernelGGL(h2e_predict_tail<FP>, g0, block, 0, stream, *k, 0u, args_dev, inst, n_instances, fc_dev);                \
        if (g_tune[3] & 16) {   /* A/B: phase 1 as one lane per instance */                                                          \
            hipLaunchKernelGGL(h2e_predict_tail<FP>, grid, block, lds_reserve, stream, *k, 1u, args_dev, inst, n_instances, fc_dev); \
        } else {                                                                                                                    \
            hipLaunchKernelGGL(h2e_predict_tail_rows<FP>, dim3((n_instances + 3) / 4), block, 0, stream, *k, args_dev, inst, n_instances); \
            hipLaunchKernelGGL(h2e_predict_tail<FP>, grid, block, 0, stream, *k, 3u, args_dev, inst, n_instances, fc_dev);           \
        }                                                                                                                           \
        hipLaunchKernelGGL(h2e_predict_tail<FP>, g2, block, 0, stream, *k, 2u, args_dev, inst, n_instances, fc_dev);                \
    } else if (phase & 1)                                                                                                           \
        hipLaunchKernelGGL(h2e_predict<FP>, grid, block, lds_reserve, stream, *k, args_dev, params_dev, aux_dev, inst, n_instances, fc_dev); \
    if ((phase & 2) && k->ecc_ops)                                                                                                  \
        hipLaunchKernelGGL(h2e_finalize_ecc<FP>, grid3, block, 0, stream, *k, inst, n_instances, fc_dev);                           \
    if ((phase & 2) && !k->ecc_ops)                                                                                                 \
        hipLaunchKernelGGL(h2e_finalize_hints<FP>, grid2, block, 0, stream, k->hint_base, n_hints, inst, n_instances, fc_dev);
    switch (field_pair) {
#if H2E_HAS_FP(0)
        case 0: { H2E_PREDICT_FP(FP_BN256_FQ) } break;
#endif
#if H2E_HAS_FP(1)
        case 1: { H2E_PREDICT_FP(FP_BLS_FQ) } break;
#endif
#if H2E_HAS_FP(2)
        case 2: { H2E_PREDICT_FP(FP_BLS_FR) } break;
#endif
        default: return -1;
    }
#undef H2E_PREDICT_FP
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Dispatchers of a split build (unit 0): the plain names the C-ABI layer calls -> the unit of the field pair
#if H2E_FP_ONLY == 0
extern "C" int h2e_engine_set_consts_fp1(int, const H2EFieldConsts*);
extern "C" int h2e_engine_set_consts_fp2(int, const H2EFieldConsts*);
extern "C" long long h2e_engine_scan_fallbacks_fp1(void);
extern "C" long long h2e_engine_scan_fallbacks_fp2(void);
extern "C" void h2e_engine_set_tuning_fp1(int, int);
extern "C" void h2e_engine_set_tuning_fp2(int, int);
extern "C" int h2e_engine_launch_fp1(int, int, const H2ELaunch*, const void*, uint32_t, const H2EFieldConsts*, hipStream_t);
extern "C" int h2e_engine_launch_fp2(int, int, const H2ELaunch*, const void*, uint32_t, const H2EFieldConsts*, hipStream_t);
extern "C" int h2e_engine_predict_fp1(int, int, const H2EPreKernel*, const uint32_t*, const uint32_t*, const uint32_t*, const void*, uint32_t,
                                      const H2EFieldConsts*, hipStream_t);
extern "C" int h2e_engine_predict_fp2(int, int, const H2EPreKernel*, const uint32_t*, const uint32_t*, const uint32_t*, const void*, uint32_t,
                                      const H2EFieldConsts*, hipStream_t);
extern "C" int h2e_engine_set_consts(int field_pair, const H2EFieldConsts* host) {
    switch (field_pair) {   // a unit's kernels only read their own pair's constants
        case 0: return h2e_engine_set_consts_fp0(field_pair, host);
        case 1: return h2e_engine_set_consts_fp1(field_pair, host);
        case 2: return h2e_engine_set_consts_fp2(field_pair, host);
        default: return -1;
    }
}
extern "C" long long h2e_engine_scan_fallbacks(void) {
    long long a = h2e_engine_scan_fallbacks_fp0(), b = h2e_engine_scan_fallbacks_fp1(), c = h2e_engine_scan_fallbacks_fp2();
    return (a < 0 || b < 0 || c < 0) ? -1 : a + b + c;
}
extern "C" void h2e_engine_set_tuning(int key, int value) {
    h2e_engine_set_tuning_fp0(key, value);
    h2e_engine_set_tuning_fp1(key, value);
    h2e_engine_set_tuning_fp2(key, value);
}
extern "C" int h2e_engine_launch(int field_pair, int mode, const H2ELaunch* launch, const void* instances, uint32_t n_instances,
                                 const H2EFieldConsts* fc_dev, hipStream_t stream) {
    switch (field_pair) {
        case 0: return h2e_engine_launch_fp0(field_pair, mode, launch, instances, n_instances, fc_dev, stream);
        case 1: return h2e_engine_launch_fp1(field_pair, mode, launch, instances, n_instances, fc_dev, stream);
        case 2: return h2e_engine_launch_fp2(field_pair, mode, launch, instances, n_instances, fc_dev, stream);
        default: return -1;
    }
}
extern "C" int h2e_engine_predict(int field_pair, int phase, const H2EPreKernel* k, const uint32_t* args_dev, const uint32_t* params_dev,
                                  const uint32_t* aux_dev, const void* instances, uint32_t n_instances, const H2EFieldConsts* fc_dev,
                                  hipStream_t stream) {
    switch (field_pair) {
        case 0: return h2e_engine_predict_fp0(field_pair, phase, k, args_dev, params_dev, aux_dev, instances, n_instances, fc_dev, stream);
        case 1: return h2e_engine_predict_fp1(field_pair, phase, k, args_dev, params_dev, aux_dev, instances, n_instances, fc_dev, stream);
        case 2: return h2e_engine_predict_fp2(field_pair, phase, k, args_dev, params_dev, aux_dev, instances, n_instances, fc_dev, stream);
        default: return -1;
    }
}
#endif
#endif   // !H2E_COLS_ON
