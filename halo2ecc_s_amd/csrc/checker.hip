// Device-side constraint check: the reference's own acceptance criterion - `MockProver::verify() == Ok`
// (src/tests/mod.rs:117-132) - evaluated on the engine's batch-interleaved advice arrays for EVERY instance of a run:
//   base gate        const + a4(next row) * next + sum a_i c_i + a0 a1 m0 + a2 a3 m1 = 0      src/circuit/base_chip.rs:50-69
//   range lookups    (tag, tagged) and (18, common) in the tagged range table                    src/circuit/range_chip.rs:119-137, table :230-258
//   range gates      one / two / three-line accumulation, switched by acc_lines                  src/circuit/range_chip.rs:141-220
//   select lookup    (value, selector 2^128 + encode, 0) in {(value, encode, is_lookup)}          src/circuit/select_chip.rs:71-88
//   copy constraints the two cells of every permutation pair hold the same value                  src/context.rs:523-541
// Unassigned advice cells and unset fixed cells evaluate to zero, as in halo2's MockProver.  Fixed cells come from the
// program's shape artefacts (dictionary ids + dictionary; constants made from instance inputs as per-instance patch values),
// advice cells from the arrays a run left in HBM, masked by the shape's `assigned` flags.
//
// This unit shares no arithmetic with the engine (engine.hip / wide_int.h): Fr is four 64-bit words with a textbook CIOS
// Montgomery product over unsigned __int128, so an error in the engine's wide-integer code cannot cancel out here - the
// constants made from instance inputs (ck_patch_values) included.  What it does share with the engine is the SHAPE: fixed
// cells, flags and the permutation list are the recorder's (compared with the oracle's by the CPU shape tests).
// Lanes: instance-minor like the arrays - consecutive lanes read consecutive instances of one cell (coalesced), the fixed
// cells of a row are the same address for every lane of that row.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/h2e.h"
#include "tape.h"

typedef uint64_t u64;
typedef uint32_t u32;
typedef unsigned __int128 u128;

namespace {
struct Fr {
    u64 v[4];
};
struct CkConsts {
    u64 n[4];
    u64 n_minv;      // -n^-1 mod 2^64
    u64 r2[4];       // R^2 mod n
    u64 shift128_m[4];   // 2^128 in Montgomery form (the select lookup's selector shift)
};
__constant__ CkConsts g_ck;

__device__ __forceinline__ bool fr_is_zero(const Fr& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
__device__ __forceinline__ bool fr_eq(const Fr& a, const Fr& b) {
    return ((a.v[0] ^ b.v[0]) | (a.v[1] ^ b.v[1]) | (a.v[2] ^ b.v[2]) | (a.v[3] ^ b.v[3])) == 0;
}
__device__ __forceinline__ bool fr_geq_n(const Fr& a) {
    for (int i = 3; i >= 0; i--) {
        if (a.v[i] > g_ck.n[i]) return true;
        if (a.v[i] < g_ck.n[i]) return false;
    }
    return true;
}
__device__ __forceinline__ Fr fr_sub_n(const Fr& a) {
    Fr r;
    u128 b = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a.v[i] - g_ck.n[i] - (u64)b;
        r.v[i] = (u64)d;
        b = (d >> 64) & 1;
    }
    return r;
}
// any 256-bit pattern -> its residue (a cell of a correct run is canonical already; a corrupted one may not be)
__device__ __forceinline__ Fr fr_canon(Fr a) {
    for (int it = 0; it < 6 && fr_geq_n(a); it++) a = fr_sub_n(a);
    return a;
}
__device__ __forceinline__ Fr fr_add(const Fr& a, const Fr& b) {   // a, b < n
    Fr r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a.v[i] + b.v[i];
        r.v[i] = (u64)c;
        c >>= 64;
    }
    return fr_geq_n(r) ? fr_sub_n(r) : r;   // (2 n < 2^256: no carry out)
}
__device__ __forceinline__ Fr fr_neg(const Fr& a) {
    if (fr_is_zero(a)) return a;
    Fr r;
    u128 b = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)g_ck.n[i] - a.v[i] - (u64)b;
        r.v[i] = (u64)d;
        b = (d >> 64) & 1;
    }
    return r;
}
// a b / R mod n (CIOS), a, b < n
__device__ Fr fr_mont_mul(const Fr& a, const Fr& b) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a.v[i] * b.v[j] + t[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (u64)c;
        t[5] = (u64)(c >> 64);
        u64 m = t[0] * g_ck.n_minv;
        c = ((u128)m * g_ck.n[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * g_ck.n[j] + t[j];
            t[j - 1] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (u64)c;
        t[4] = t[5] + (u64)(c >> 64);
    }
    Fr r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fr_geq_n(r)) r = fr_sub_n(r);
    return r;
}
__device__ __forceinline__ Fr fr_to_mont(const Fr& a) {
    Fr r2 = {{g_ck.r2[0], g_ck.r2[1], g_ck.r2[2], g_ck.r2[3]}};
    return fr_mont_mul(a, r2);
}
// plain product a b mod n of canonical values: (a R)(b) / R
__device__ __forceinline__ Fr fr_mul(const Fr& a, const Fr& b) { return fr_mont_mul(fr_to_mont(a), b); }

struct Region {
    const u64* adv;        // batch-interleaved [row][cols][2][n_inst][2 words]
    const uint8_t* flags;  // [row][cols] bit 0 = assigned
    const u32* fix;        // [row][fcols] dictionary ids (0 = unset); base: bit 31 set = per-instance patch value, index in the low bits
    u64 rows;              // rows the arrays hold
    u64 height;            // rows the gates run over
};
template <int COLS>
__device__ __forceinline__ Fr ld_adv(const Region& R, u64 row, int col, u32 inst, u32 n_inst) {
    Fr r = {{0, 0, 0, 0}};
    if (row >= R.rows || !(R.flags[row * COLS + col] & 1)) return r;
    const u64* p = R.adv + ((row * COLS + col) * 2 * (u64)n_inst + inst) * 2;
    r.v[0] = p[0];
    r.v[1] = p[1];
    r.v[2] = p[2 * (u64)n_inst];
    r.v[3] = p[2 * (u64)n_inst + 1];
    return fr_canon(r);
}
__device__ __forceinline__ Fr ld_dict(const u64* dict, u32 id) {
    Fr r = {{dict[4 * (u64)id], dict[4 * (u64)id + 1], dict[4 * (u64)id + 2], dict[4 * (u64)id + 3]}};
    return r;
}
// fail[inst][class] counts, fail[inst][CLASSES + class] = the lowest failing row (pair index for copy constraints)
__device__ __forceinline__ void note_fail(u64* fail, u32 inst, int cls, u64 where) {
    atomicAdd((unsigned long long*)(fail + (size_t)inst * 2 * H2E_CHECK_CLASSES + cls), 1ull);
    atomicMin((unsigned long long*)(fail + (size_t)inst * 2 * H2E_CHECK_CLASSES + H2E_CHECK_CLASSES + cls), (unsigned long long)where);
}

// ---- base gate (base_chip.rs:50-69) ------------------------------------------------------------------------------------
// fixed columns [c0..c4, m0, m1, next, const] (context.rs:367-375); dict_m = the dictionary in Montgomery form, so that
// mont_mul(advice, coefficient) is their plain product
__global__ void __launch_bounds__(256) ck_base_gate(Region B, const u64* __restrict__ dict, const u64* __restrict__ dict_m,
                                                    const u64* __restrict__ patch_vals, u32 n_patches, u32 n_inst, u64* __restrict__ fail) {
    u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    u64 row = t / n_inst;
    u32 inst = (u32)(t % n_inst);
    if (row >= B.height) return;
    const u32* ids = B.fix + row * 9;
    auto coeff_m = [&](int k) -> u32 { return ids[k]; };
    Fr acc = {{0, 0, 0, 0}};
    {   // const (may be made from this instance's inputs)
        u32 id = ids[8];
        if (id & 0x80000000u) {
            const u64* pv = patch_vals + ((size_t)inst * n_patches + (id & 0x7fffffffu)) * 4;
            Fr c = {{pv[0], pv[1], pv[2], pv[3]}};
            acc = c;
        } else if (id) {
            acc = ld_dict(dict, id);
        }
    }
    {   // a4 of the next row * next
        u32 id = coeff_m(7);
        if (id) acc = fr_add(acc, fr_mont_mul(ld_adv<5>(B, row + 1, 4, inst, n_inst), ld_dict(dict_m, id)));
    }
    Fr a[5];
    bool have[5] = {false, false, false, false, false};
    auto adv = [&](int i) -> const Fr& {
        if (!have[i]) {
            a[i] = ld_adv<5>(B, row, i, inst, n_inst);
            have[i] = true;
        }
        return a[i];
    };
    for (int i = 0; i < 5; i++) {
        u32 id = coeff_m(i);
        if (id) acc = fr_add(acc, fr_mont_mul(adv(i), ld_dict(dict_m, id)));
    }
    for (int i = 0; i < 2; i++) {
        u32 id = coeff_m(5 + i);
        if (id) {
            Fr p = fr_mul(adv(2 * i), adv(2 * i + 1));
            acc = fr_add(acc, fr_mont_mul(p, ld_dict(dict_m, id)));
        }
    }
    if (!fr_is_zero(acc)) note_fail(fail, inst, H2E_CHECK_BASE_GATE, row);
}

// ---- range chip (range_chip.rs:119-220; table :230-258: tag in 0..=18, value < 2^tag) -------------------------------------
// advice [acc, tagged, common], fixed [acc_lines, tag] (range_chip.rs:81-92)
__device__ __forceinline__ bool small_below(const Fr& x, u64 bits) {   // x < 2^bits, bits <= 18
    return (x.v[1] | x.v[2] | x.v[3]) == 0 && x.v[0] < (1ull << bits);
}
__global__ void __launch_bounds__(256) ck_range(Region G, const u64* __restrict__ dict, const u64* __restrict__ shifts_m, u32 n_inst,
                                                u64* __restrict__ fail) {
    u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    u64 row = t / n_inst;
    u32 inst = (u32)(t % n_inst);
    if (row >= G.height) return;
    u32 lines_id = row < G.rows ? G.fix[row * 2] : 0, tag_id = row < G.rows ? G.fix[row * 2 + 1] : 0;
    Fr tagged = ld_adv<3>(G, row, 1, inst, n_inst), common = ld_adv<3>(G, row, 2, inst, n_inst);
    {   // the two lookups
        Fr tag = {{0, 0, 0, 0}};
        if (tag_id) tag = ld_dict(dict, tag_id);
        bool ok = small_below(tag, 5) && tag.v[0] <= 18 && small_below(tagged, tag.v[0]);
        bool ok2 = small_below(common, 18);
        if (!ok || !ok2) note_fail(fail, inst, H2E_CHECK_RANGE_LOOKUP, row);
    }
    if (!lines_id) return;   // acc_lines = 0: every gate is multiplied by it
    Fr lines = ld_dict(dict, lines_id);
    if (fr_is_zero(lines)) return;
    bool is_small = (lines.v[1] | lines.v[2] | lines.v[3]) == 0 && lines.v[0] <= 3;
    Fr acc_v = ld_adv<3>(G, row, 0, inst, n_inst);
    // gate nl is  (acc - sum) * lines * prod_{root != nl} (lines - root): for lines in {1, 2, 3} only gate `lines` is live; any
    // other non-zero value of the fixed cell leaves all three live
    for (u32 nl = 1; nl <= 3; nl++) {
        if (is_small && lines.v[0] != nl) continue;
        Fr sum = {{0, 0, 0, 0}};
        if (nl == 1) {
            sum = tagged;
        } else {
            // shifts_m[k] = 2^(18 k) in Montgomery form: common cells of the nl rows, then their tagged cells
            for (u32 j = 0; j < nl; j++) {
                Fr c = j == 0 ? common : ld_adv<3>(G, row + j, 2, inst, n_inst);
                Fr s = {{shifts_m[4 * j], shifts_m[4 * j + 1], shifts_m[4 * j + 2], shifts_m[4 * j + 3]}};
                sum = fr_add(sum, fr_mont_mul(c, s));
            }
            for (u32 j = 0; j < nl; j++) {
                Fr c = j == 0 ? tagged : ld_adv<3>(G, row + j, 1, inst, n_inst);
                const u64* sp = shifts_m + 4 * (nl + j);
                Fr s = {{sp[0], sp[1], sp[2], sp[3]}};
                sum = fr_add(sum, fr_mont_mul(c, s));
            }
        }
        if (!fr_eq(acc_v, sum)) note_fail(fail, inst, H2E_CHECK_RANGE_GATE, row);
    }
}

// ---- select chip lookup_any (select_chip.rs:71-88) -------------------------------------------------------------------------
// advice [value, selector], fixed [encode, is_lookup].  Table = the rows whose is_lookup is zero, as (value, encode) - their
// encode cells are fixed, so the host sorts them once per shape: keys[k] = encode (canonical words, compared from the top
// word down), key_rows[k] = the row.  Every row looks up (value, selector 2^128 + encode): a table row finds itself; a
// `get` row must find a `set` row with its encode and the same value.
__device__ __forceinline__ int key_cmp(const u64* a, const Fr& b) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] < b.v[i]) return -1;
        if (a[i] > b.v[i]) return 1;
    }
    return 0;
}
__global__ void __launch_bounds__(256) ck_select(Region S, const u64* __restrict__ dict, const u64* __restrict__ keys, const u32* __restrict__ key_rows,
                                                 u32 n_keys, u32 n_inst, u64* __restrict__ fail) {
    u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    u64 row = t / n_inst;
    u32 inst = (u32)(t % n_inst);
    if (row >= S.height) return;
    u32 enc_id = row < S.rows ? S.fix[row * 2] : 0, look_id = row < S.rows ? S.fix[row * 2 + 1] : 0;
    bool is_lookup = look_id && !fr_is_zero(ld_dict(dict, look_id));
    Fr sel = ld_adv<2>(S, row, 1, inst, n_inst);
    if (!is_lookup && fr_is_zero(sel)) return;   // the row is its own table entry
    Fr value = ld_adv<2>(S, row, 0, inst, n_inst);
    Fr enc = {{0, 0, 0, 0}};
    if (enc_id) enc = ld_dict(dict, enc_id);
    Fr sh = {{g_ck.shift128_m[0], g_ck.shift128_m[1], g_ck.shift128_m[2], g_ck.shift128_m[3]}};
    enc = fr_add(enc, fr_mont_mul(sel, sh));
    // first table entry with that encode, then every entry with it
    u32 lo = 0, hi = n_keys;
    while (lo < hi) {
        u32 mid = (lo + hi) / 2;
        if (key_cmp(keys + 4 * (u64)mid, enc) < 0) lo = mid + 1;
        else hi = mid;
    }
    bool found = false;
    for (u32 k = lo; k < n_keys && !found && key_cmp(keys + 4 * (u64)k, enc) == 0; k++) {
        u32 trow = key_rows[k];
        Fr tv = trow == 0xffffffffu ? Fr{{0, 0, 0, 0}} : ld_adv<2>(S, trow, 0, inst, n_inst);   // (0xffffffff: the all-zero row of the unused part of the circuit)
        found = fr_eq(tv, value);
    }
    if (!found) note_fail(fail, inst, H2E_CHECK_SELECT_LOOKUP, row);
}

// ---- copy constraints (context.rs:523-541) ---------------------------------------------------------------------------------
struct Regions3 {
    Region r[3];
};
__device__ __forceinline__ bool ld_ref(const Regions3& A, u32 ref, u32 inst, u32 n_inst, Fr& out) {
    u32 region = H2E_REF_REGION(ref), col = H2E_REF_COL(ref);
    u64 row = H2E_REF_ROW(ref);
    const int cols = region == 0 ? 5 : region == 1 ? 3 : 2;
    if (region > 2 || (int)col >= cols) return false;
    const Region& R = A.r[region];
    if (row >= R.rows) return false;
    uint8_t f = R.flags[row * cols + col];
    const u64* p = R.adv + ((row * cols + col) * 2 * (u64)n_inst + inst) * 2;
    out.v[0] = p[0];
    out.v[1] = p[1];
    out.v[2] = p[2 * (u64)n_inst];
    out.v[3] = p[2 * (u64)n_inst + 1];
    return (f & 3) == 3;   // assigned and enabled for the permutation argument
}
__global__ void __launch_bounds__(256) ck_copy(Regions3 A, const u32* __restrict__ perms, u64 n_pairs, u32 n_inst, u64* __restrict__ fail) {
    u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    u64 k = t / n_inst;
    u32 inst = (u32)(t % n_inst);
    if (k >= n_pairs) return;
    Fr a, b;
    bool ok = ld_ref(A, perms[2 * k], inst, n_inst, a);
    ok = ld_ref(A, perms[2 * k + 1], inst, n_inst, b) && ok;
    if (!ok || !fr_eq(a, b)) note_fail(fail, inst, H2E_CHECK_COPY, k);
}

__global__ void ck_to_mont(const u64* __restrict__ in, u64* __restrict__ out, u64 n) {
    u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Fr x = {{in[4 * k], in[4 * k + 1], in[4 * k + 2], in[4 * k + 3]}};
    Fr y = fr_to_mont(fr_canon(x));
    for (int i = 0; i < 4; i++) out[4 * k + i] = y.v[i];
}
__global__ void ck_init_fail(u64* fail, u32 n_inst) {
    u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_inst * 2 * H2E_CHECK_CLASSES) return;
    fail[k] = (k % (2 * H2E_CHECK_CLASSES)) < H2E_CHECK_CLASSES ? 0ull : ~0ull;
}
}   // namespace

struct H2ECheckRegion {   // (mirrors Region; plain C for the C-ABI layer)
    const void* adv;
    const uint8_t* flags;
    const uint32_t* fix;
    uint64_t rows, height;
};
extern "C" int h2e_engine_check_consts(const uint64_t n[4], uint64_t n_minv, const uint64_t r2[4]) {
    CkConsts c;
    for (int i = 0; i < 4; i++) {
        c.n[i] = n[i];
        c.r2[i] = r2[i];
    }
    c.n_minv = n_minv;
    // 2^128 in Montgomery form = 2^128 R mod n = mont_mul(2^128, R^2): on the host, with the same CIOS over __int128
    u64 a[4] = {0, 0, 1, 0}, t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 cy = 0;
        for (int j = 0; j < 4; j++) {
            cy += (u128)a[i] * r2[j] + t[j];
            t[j] = (u64)cy;
            cy >>= 64;
        }
        cy += t[4];
        t[4] = (u64)cy;
        t[5] = (u64)(cy >> 64);
        u64 m = t[0] * n_minv;
        cy = ((u128)m * n[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) {
            cy += (u128)m * n[j] + t[j];
            t[j - 1] = (u64)cy;
            cy >>= 64;
        }
        cy += t[4];
        t[3] = (u64)cy;
        t[4] = t[5] + (u64)(cy >> 64);
    }
    bool ge = t[4] != 0;
    if (!ge) {
        ge = true;
        for (int i = 3; i >= 0; i--) {
            if (t[i] > n[i]) break;
            if (t[i] < n[i]) { ge = false; break; }
        }
    }
    if (ge) {
        u128 b = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)t[i] - n[i] - (u64)b;
            t[i] = (u64)d;
            b = (d >> 64) & 1;
        }
    }
    for (int i = 0; i < 4; i++) c.shift128_m[i] = t[i];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ck), &c, sizeof(c));
}
extern "C" int h2e_engine_check_to_mont(const uint64_t* in, uint64_t* out, uint64_t n, hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(ck_to_mont, dim3((u32)((n + 255) / 256)), dim3(256), 0, stream, in, out, n);
    return (int)hipGetLastError();
}
// The base gate's constants that are made from instance inputs (the G2 coordinates of a pairing check: limbs and natives of
// assign_int_constant, src/circuit/integer_chip.rs:580-598), computed HERE from the input words - not taken from the engine's
// h2e_fixed_patches - so that an error in the engine's limb split / mod n cannot cancel out: patch = [base row, fixed column,
// input slot, limb], limb >= 0: bits [108 limb, 108 limb + 108) of the slot's value, limb = -1: the value mod n (Horner over its
// 64-bit words in this unit's Montgomery arithmetic).  out: [instance][patch][4 words], canonical.
__global__ void __launch_bounds__(256) ck_patch_values(const u32* __restrict__ patches, u32 n_patches, const u64* __restrict__ inputs, u32 n_slots,
                                                       u32 slot_words, u32 n_inst, u64* __restrict__ out) {
    u32 t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_patches * n_inst) return;
    u32 inst = t / n_patches, k = t % n_patches;
    const u32 slot = patches[4 * k + 2];
    const int limb = (int)patches[4 * k + 3];
    const u64* x = inputs + ((size_t)inst * n_slots + slot) * slot_words;
    Fr v = {{0, 0, 0, 0}};
    if (limb >= 0) {
        auto bit = [&](u32 b) -> u64 { return b / 64 < slot_words ? (x[b / 64] >> (b % 64)) & 1ull : 0ull; };
        for (u32 b = 0; b < 108; b++) v.v[b / 64] |= bit(108u * (u32)limb + b) << (b % 64);   // (bit by bit: nothing to share with anybody's shift code)
    } else {
        Fr two64 = {{0, 1, 0, 0}};
        const Fr two64_m = fr_to_mont(two64);
        Fr acc = {{0, 0, 0, 0}};                       // Montgomery form
        for (int w = (int)slot_words - 1; w >= 0; w--) {
            Fr word = {{x[w], 0, 0, 0}};
            acc = fr_add(fr_mont_mul(acc, two64_m), fr_to_mont(word));
        }
        Fr one = {{1, 0, 0, 0}};
        v = fr_mont_mul(acc, one);                     // out of Montgomery form
    }
    u64* o = out + ((size_t)inst * n_patches + k) * 4;
    for (int i = 0; i < 4; i++) o[i] = v.v[i];
}
extern "C" int h2e_engine_check_patch_values(const uint32_t* patches, uint32_t n_patches, const uint64_t* inputs, uint32_t n_slots, uint32_t slot_words,
                                             uint32_t n_instances, uint64_t* out, hipStream_t stream) {
    if (n_patches == 0 || n_instances == 0) return 0;
    hipLaunchKernelGGL(ck_patch_values, dim3((n_patches * n_instances + 255) / 256), dim3(256), 0, stream, patches, n_patches, inputs, n_slots, slot_words,
                       n_instances, out);
    return (int)hipGetLastError();
}
// classes: bit k of `classes` = run check class k (H2E_CHECK_*; the two range classes run together)
extern "C" int h2e_engine_check(const H2ECheckRegion* regs, const uint64_t* dict, const uint64_t* dict_m, const uint64_t* shifts_m,
                                const uint64_t* patch_vals, uint32_t n_patches, const uint64_t* sel_keys, const uint32_t* sel_key_rows,
                                uint32_t n_sel_keys, const uint32_t* perms, uint64_t n_pairs, uint32_t n_instances, uint32_t classes,
                                uint64_t* fail, hipStream_t stream) {
    if (n_instances == 0) return 0;
    Region R[3];
    for (int i = 0; i < 3; i++) {
        R[i].adv = (const u64*)regs[i].adv;
        R[i].flags = regs[i].flags;
        R[i].fix = regs[i].fix;
        R[i].rows = regs[i].rows;
        R[i].height = regs[i].height;
    }
    auto blocks = [&](u64 items) -> u64 { return (items * n_instances + 255) / 256; };
    for (int i = 0; i < 3; i++)
        if (blocks(R[i].height) > 0x7fffffffull) return -1;
    if (blocks(n_pairs) > 0x7fffffffull) return -1;
    hipLaunchKernelGGL(ck_init_fail, dim3((n_instances * 2 * H2E_CHECK_CLASSES + 255) / 256), dim3(256), 0, stream, fail, n_instances);
    if ((classes & (1u << H2E_CHECK_BASE_GATE)) && R[0].height)
        hipLaunchKernelGGL(ck_base_gate, dim3((u32)blocks(R[0].height)), dim3(256), 0, stream, R[0], dict, dict_m, patch_vals, n_patches, n_instances, fail);
    if ((classes & ((1u << H2E_CHECK_RANGE_GATE) | (1u << H2E_CHECK_RANGE_LOOKUP))) && R[1].height)
        hipLaunchKernelGGL(ck_range, dim3((u32)blocks(R[1].height)), dim3(256), 0, stream, R[1], dict, shifts_m, n_instances, fail);
    if ((classes & (1u << H2E_CHECK_SELECT_LOOKUP)) && R[2].height)
        hipLaunchKernelGGL(ck_select, dim3((u32)blocks(R[2].height)), dim3(256), 0, stream, R[2], dict, sel_keys, sel_key_rows, n_sel_keys, n_instances, fail);
    if ((classes & (1u << H2E_CHECK_COPY)) && n_pairs) {
        Regions3 A;
        for (int i = 0; i < 3; i++) A.r[i] = R[i];
        hipLaunchKernelGGL(ck_copy, dim3((u32)blocks(n_pairs)), dim3(256), 0, stream, A, perms, n_pairs, n_instances, fail);
    }
    return (int)hipGetLastError();
}
