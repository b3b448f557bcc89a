// The operator API: a device-resident Context (h2e_records) whose chip ops are recorded and run one by one, with the op-program cache.
// Included by h2e_capi.cpp INSIDE its extern "C" block.
// =================================================================================================================
// Operator API: a device-resident Context (include/h2e.h "operator API").  The reference's operator surface is a
// Context you call chip ops on; its own seam for running part of the work elsewhere is fork-at-offset / merge
// (ParallelClone, src/circuit/ecc_chip.rs:64-77).  h2e_records is that Context for a batch of instances: advice arrays
// in HBM, cursors / heights / msm prefix / shape artefacts on the host.  Every op records one program that starts at the
// current offsets with its operands as handles to earlier rows, runs it on the shared arrays and advances the cursors.
struct h2e_records {
    h2e_ctx* ctx = nullptr;
    int field_pair = 0, scalar_field = -1;
    uint32_t n_instances = 0;
    bool emit_shape = true, own_arrays = false;
    uint64_t cap[3] = {0, 0, 0};
    void* d_arr[3] = {nullptr, nullptr, nullptr};
    uint32_t* d_status = nullptr;
    // Context (src/context.rs:40-46) + Records heights (:297-299) + NativeScalarEccContext.1 / msm_prefix
    uint64_t off[3] = {0, 0, 0}, height[3] = {0, 0, 0};
    size_t msm_prefix = 0;
    uint64_t n_advice_cells = 0, n_ops = 0;
    // accumulated shape artefacts
    std::vector<h2e::FrVal> dict;
    std::unordered_map<h2e::FrVal, uint32_t, h2e::FrValHash> dict_map;
    std::vector<uint32_t> fix[3];
    std::vector<uint8_t> flags[3];
    std::vector<uint32_t> perm_flat;
    std::vector<uint32_t> patch_flat;   // [row, fixed col, op index << 16 | input slot, limb]
    void* d_dummy = nullptr;            // input vector of ops that take none (the engine wants a valid pointer)
    h2e_records() { dict.push_back(h2e::FrVal{0, 0, 0, 0}); }
    ~h2e_records() {
        if (ctx) {
            (void)hipSetDevice(ctx->device);
            (void)hipDeviceSynchronize();
        }
        (void)hipFree(d_dummy);
        if (own_arrays) {
            for (int i = 0; i < 3; i++) (void)hipFree(d_arr[i]);
            (void)hipFree(d_status);
        }
    }
    uint32_t intern(const h2e::FrVal& v) {
        auto it = dict_map.find(v);
        if (it != dict_map.end()) return it->second;
        uint32_t id = (uint32_t)dict.size();
        dict.push_back(v);
        dict_map.emplace(v, id);
        return id;
    }
};

namespace {
const int FIXC[3] = {9, 2, 2}, ADVC[3] = {5, 3, 2};
static h2e::AssignedInteger to_int(const h2e_int& a) {
    h2e::AssignedInteger r;
    for (int i = 0; i < H2E_MAX_L; i++) r.limbs_le[i] = a.limbs[i];
    r.native = a.native;
    r.times = a.times;
    return r;
}
h2e_int from_int(const h2e::AssignedInteger& a) {
    h2e_int r;
    for (int i = 0; i < H2E_MAX_L; i++) r.limbs[i] = a.limbs_le[i];
    r.native = a.native;
    r.times = (uint32_t)a.times;
    return r;
}
static h2e::AssignedPoint to_point(const h2e_point& p) { return h2e::AssignedPoint{to_int(p.x), to_int(p.y), h2e::AssignedCondition{h2e::AssignedValue{p.z}}}; }
h2e_point from_point(const h2e::AssignedPoint& p) {
    h2e_point r;
    r.x = from_int(p.x);
    r.y = from_int(p.y);
    r.z = p.z.v.ref;
    return r;
}

// Record one op at the records' current state (or take its program from the context's op cache), run it, merge its shape
// artefacts and advance the Context.  `key` names the op and everything its recording depends on besides the records' state;
// `outs` lists the caller's output handles (filled by `body` when the op is recorded, from the cache otherwise).
struct OpOut {
    void* ptr;
    size_t bytes;
};
static std::string key_of(const char* name, std::initializer_list<std::pair<const void*, size_t>> blobs) {
    std::string k(name);
    for (auto& b : blobs) {
        k.push_back('|');
        if (b.first) k.append((const char*)b.first, b.second);
    }
    return k;
}
int records_op(h2e_records* R, const std::string& key, uint32_t n_slots, const void* d_inputs, void* stream, std::initializer_list<OpOut> outs,
               const std::function<void(h2e::Recorder&, h2e::NativeScalarEccContext&, uint32_t)>& body) {
    if (!R) return fail(H2E_ERR_INVALID, "null records");
    if (n_slots && !d_inputs) return fail(H2E_ERR_INVALID, "the op takes inputs: d_inputs is null");
    if (R->n_ops >= 65535) return fail(H2E_ERR_SHAPE, "records: more than 65535 ops (the fixed-patch list packs the op index in 16 bits)");
    h2e_ctx* ctx = R->ctx;
    std::string full = key;
    {
        uint64_t st[9] = {R->off[0], R->off[1], R->off[2], R->height[0], R->height[1], R->height[2], (uint64_t)R->msm_prefix,
                          (uint64_t)(R->field_pair * 8 + (R->scalar_field + 1)), (uint64_t)(R->emit_shape ? 1 : 0) | ((uint64_t)n_slots << 8)};
        full.push_back('#');
        full.append((const char*)st, sizeof(st));
    }
    h2e_program* p = nullptr;
    size_t prefix_after = R->msm_prefix;   // (applied when the op has run: a failing op leaves the records' state as it was)
    struct Release {   // the entry cannot be evicted while this call runs its program
        h2e_ctx* ctx;
        const std::string* key;
        bool held = false;
        ~Release() {
            if (!held) return;
            std::lock_guard<std::mutex> g(ctx->op_mu);
            auto it = ctx->op_cache.find(*key);
            if (it != ctx->op_cache.end() && it->second.in_use) it->second.in_use--;
        }
    } release{ctx, &full};
    {
        std::lock_guard<std::mutex> g(ctx->op_mu);
        auto it = ctx->op_cache.find(full);
        if (it != ctx->op_cache.end()) {
            p = it->second.prog;
            size_t k = 0;
            for (auto& o : outs) {
                if (o.ptr && k < it->second.outs.size() && it->second.outs[k].size() == o.bytes) std::memcpy(o.ptr, it->second.outs[k].data(), o.bytes);
                k++;
            }
            prefix_after = it->second.msm_prefix_after;
            it->second.last_use = ++ctx->op_tick;
            it->second.in_use++;
            release.held = true;
            ctx->op_hits++;
        }
    }
    if (!p) {
        std::unique_ptr<h2e_program> np(new h2e_program());
        np->field_pair = R->field_pair;
        try {
            np->rec.reset(new h2e::Recorder(field_pair(R->field_pair)));
            h2e::Recorder& r = *np->rec;
            r.emit_shape = R->emit_shape;
            // clone_with_offset of the caller's context (context.rs:145-158): cursors and heights carry over
            r.base_offset = R->off[0];
            r.range_offset = R->off[1];
            r.select_offset = R->off[2];
            r.base_height = R->height[0];
            r.range_height = R->height[1];
            r.select_height = R->height[2];
            h2e::NativeScalarEccContext ecc(r, R->field_pair == H2E_FIELD_BN256_FQ ? h2e::bn256_g1_params() : h2e::bls12_381_g1_params(), R->msm_prefix);
            ecc.scalar_field = R->scalar_field;
            ecc.with_select = ecc.has_select_chip();
            uint32_t s0 = r.alloc_inputs(std::max<uint32_t>(1, n_slots));
            body(r, ecc, s0);
            np->finish();
            prefix_after = ecc.msm_prefix;
        } catch (std::exception& e) {
            return fail(H2E_ERR_SHAPE, e.what());
        }
        if (np->base_rows > R->cap[0] || np->range_rows > R->cap[1] || np->select_rows > R->cap[2])
            return fail(H2E_ERR_SHAPE, "records: the op does not fit the arrays' capacity (like HALO2ECC_S_MAX_ROWS, src/context.rs:36)");
        std::lock_guard<std::mutex> g(ctx->op_mu);
        h2e_ctx::OpEntry& e = ctx->op_cache[full];
        if (!e.prog) {
            e.prog = np.release();
            for (auto& o : outs) e.outs.emplace_back((const uint8_t*)o.ptr, (const uint8_t*)o.ptr + (o.ptr ? o.bytes : 0));
            e.msm_prefix_after = prefix_after;
            ctx->op_misses++;
        }
        e.last_use = ++ctx->op_tick;
        e.in_use++;
        release.held = true;
        p = e.prog;
        ctx->op_cache_trim();
    }
    if (p->base_rows > R->cap[0] || p->range_rows > R->cap[1] || p->select_rows > R->cap[2])
        return fail(H2E_ERR_SHAPE, "records: the op does not fit the arrays' capacity (like HALO2ECC_S_MAX_ROWS, src/context.rs:36)");
    h2e::Recorder& r = *p->rec;
    int rc = 0;
    if (!r.tape.empty()) {
        const void* in = d_inputs;
        if (!in) {   // ops without inputs still get a valid (unused) pointer
            if (!R->d_dummy) {
                HIP_TRY(hipSetDevice(ctx->device));
                HIP_TRY(hipMalloc(&R->d_dummy, (size_t)R->n_instances * 64));
                HIP_TRY(hipMemset(R->d_dummy, 0, (size_t)R->n_instances * 64));
            }
            in = R->d_dummy;
        }
        rc = h2e_run(ctx, p, R->n_instances, in, R->d_arr[0], R->d_arr[1], R->d_arr[2], R->d_status, stream);
    }
    if (rc) return rc;
    R->msm_prefix = prefix_after;
    // merge (ParallelClone::merge + apply_offset_diff)
    uint64_t before[3] = {R->off[0], R->off[1], R->off[2]};
    R->off[0] = r.base_offset;
    R->off[1] = r.range_offset;
    R->off[2] = r.select_offset;
    R->height[0] = r.base_height;
    R->height[1] = r.range_height;
    R->height[2] = r.select_height;
    if (R->emit_shape) {
        const std::vector<uint32_t>* pfix[3] = {&r.base_fix, &r.range_fix, &r.select_fix};
        const std::vector<uint8_t>* pfl[3] = {&r.base_flags, &r.range_flags, &r.select_flags};
        uint64_t rows[3] = {p->base_rows, p->range_rows, p->select_rows};
        std::vector<uint32_t> idmap(r.dict.size(), 0);
        for (size_t i = 1; i < r.dict.size(); i++) idmap[i] = R->intern(r.dict[i]);
        for (int reg = 0; reg < 3; reg++) {
            if (R->fix[reg].size() < rows[reg] * FIXC[reg]) R->fix[reg].resize(rows[reg] * FIXC[reg], 0);
            if (R->flags[reg].size() < rows[reg] * ADVC[reg]) R->flags[reg].resize(rows[reg] * ADVC[reg], 0);
            for (uint64_t row = before[reg]; row < rows[reg]; row++) {
                for (int c = 0; c < FIXC[reg]; c++) {
                    size_t k = row * FIXC[reg] + c;
                    if (k < pfix[reg]->size() && (*pfix[reg])[k]) R->fix[reg][k] = idmap[(*pfix[reg])[k]];
                }
                for (int c = 0; c < ADVC[reg]; c++) {
                    size_t k = row * ADVC[reg] + c;
                    if (k < pfl[reg]->size()) R->flags[reg][k] |= (*pfl[reg])[k];
                }
            }
        }
        auto set_perm = [&](uint32_t cell) {
            uint32_t reg = H2E_REF_REGION(cell);
            size_t k = (size_t)H2E_REF_ROW(cell) * ADVC[reg] + H2E_REF_COL(cell);
            if (R->flags[reg].size() <= k) R->flags[reg].resize(k + 1, 0);
            R->flags[reg][k] |= 2;
        };
        for (auto& pr : r.permutations) {
            R->perm_flat.push_back(pr.first);
            R->perm_flat.push_back(pr.second);
            set_perm(pr.first);
            set_perm(pr.second);
        }
        for (auto& fpch : r.fixed_patches) {
            R->patch_flat.push_back(fpch.row);
            R->patch_flat.push_back(fpch.col);
            R->patch_flat.push_back((uint32_t)(R->n_ops << 16) | fpch.input_slot);
            R->patch_flat.push_back((uint32_t)fpch.limb);
        }
        R->n_advice_cells += r.n_advice_cells;
    }
    R->n_ops++;
    return 0;
}
}  // namespace

int h2e_records_create(h2e_ctx* ctx, int field_pair_id, int scalar_field, uint32_t n_instances, uint64_t base_rows, uint64_t range_rows,
                       uint64_t select_rows, int emit_shape, h2e_records** out) {
    if (!ctx || !out) return fail(H2E_ERR_INVALID, "null argument");
    if (field_pair_id < 0 || field_pair_id > 2 || scalar_field < -1 || scalar_field > 2) return fail(H2E_ERR_INVALID, "bad field pair");
    if (n_instances == 0 || base_rows == 0 || range_rows == 0 || select_rows == 0) return fail(H2E_ERR_INVALID, "empty records");
    // a flag word, not a boolean: a caller's "true" of 2 or -1 must not silently mean "no shape" / "no select chip"
    if (emit_shape & ~(H2E_RECORDS_EMIT_SHAPE | H2E_RECORDS_NO_SELECT_CHIP)) return fail(H2E_ERR_INVALID, "h2e_records_create: unknown bits in the flag word (H2E_RECORDS_*)");
    h2e_records* R = new h2e_records();
    R->ctx = ctx;
    R->field_pair = field_pair_id;
    R->scalar_field = scalar_field;
    R->n_instances = n_instances;
    R->emit_shape = (emit_shape & H2E_RECORDS_EMIT_SHAPE) != 0;
    // NativeScalarEccContext::new_without_select_chip (src/context.rs:201-205): the msm prefix is usize::MAX and msm_unsafe takes
    // the bisection form (src/circuit/ecc_chip.rs:373-408 dispatches on has_select_chip, native_scalar_ecc_chip.rs:27-46)
    if (emit_shape & H2E_RECORDS_NO_SELECT_CHIP) R->msm_prefix = (size_t)-1;
    R->cap[0] = base_rows;
    R->cap[1] = range_rows;
    R->cap[2] = select_rows;
    R->own_arrays = true;
    hipError_t e = hipSetDevice(ctx->device);
    for (int i = 0; i < 3 && e == hipSuccess; i++) {
        size_t bytes = (size_t)R->cap[i] * ADVC[i] * 32 * n_instances;
        e = hipMalloc(&R->d_arr[i], bytes);
        if (e == hipSuccess) e = hipMemset(R->d_arr[i], 0, bytes);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&R->d_status, (size_t)n_instances * 4);
    if (e == hipSuccess) e = hipMemset(R->d_status, 0, (size_t)n_instances * 4);
    if (e != hipSuccess) {
        delete R;
        return fail(H2E_ERR_HIP, std::string("records: ") + hipGetErrorString(e));
    }
    *out = R;
    return 0;
}
int h2e_records_attach(h2e_ctx* ctx, int field_pair_id, int scalar_field, uint32_t n_instances, void* d_base, void* d_range, void* d_select,
                       void* d_status, const uint64_t capacity_rows[3], const uint64_t offset0[3], uint64_t msm_prefix0, int emit_shape,
                       h2e_records** out) {
    if (!ctx || !out || !capacity_rows || !offset0) return fail(H2E_ERR_INVALID, "null argument");
    if (!d_base || !d_range || !d_select || !d_status) return fail(H2E_ERR_INVALID, "null device pointer");
    if (field_pair_id < 0 || field_pair_id > 2 || scalar_field < -1 || scalar_field > 2) return fail(H2E_ERR_INVALID, "bad field pair");
    if (n_instances == 0) return fail(H2E_ERR_INVALID, "empty records");
    for (int i = 0; i < 3; i++)
        if (capacity_rows[i] == 0 || offset0[i] >= capacity_rows[i] || capacity_rows[i] > (1ull << 26))
            return fail(H2E_ERR_INVALID, "offsets must lie inside the arrays (at most 2^26 rows)");
    h2e_records* R = new h2e_records();
    R->ctx = ctx;
    R->field_pair = field_pair_id;
    R->scalar_field = scalar_field;
    R->n_instances = n_instances;
    R->emit_shape = emit_shape != 0;
    R->own_arrays = false;
    R->d_arr[0] = d_base;
    R->d_arr[1] = d_range;
    R->d_arr[2] = d_select;
    R->d_status = (uint32_t*)d_status;
    for (int i = 0; i < 3; i++) {
        R->cap[i] = capacity_rows[i];
        R->off[i] = offset0[i];
        R->height[i] = offset0[i];
    }
    R->msm_prefix = (size_t)msm_prefix0;
    *out = R;
    return 0;
}
void h2e_records_destroy(h2e_records* R) { delete R; }
int h2e_records_arrays(h2e_records* R, void** d_base, void** d_range, void** d_select, void** d_status) {
    if (!R) return fail(H2E_ERR_INVALID, "null records");
    if (d_base) *d_base = R->d_arr[0];
    if (d_range) *d_range = R->d_arr[1];
    if (d_select) *d_select = R->d_arr[2];
    if (d_status) *d_status = R->d_status;
    return 0;
}
int h2e_records_shape(const h2e_records* R, h2e_shape* out) {
    if (!R || !out) return fail(H2E_ERR_INVALID, "null argument");
    std::memset(out, 0, sizeof(*out));
    out->field_pair = R->field_pair;
    out->slot_words = field_pair(R->field_pair).w_words;
    out->base_offset = R->off[0];
    out->range_offset = R->off[1];
    out->select_offset = R->off[2];
    out->base_height = R->height[0];
    out->range_height = R->height[1];
    out->select_height = R->height[2];
    out->base_rows = R->cap[0];
    out->range_rows = R->cap[1];
    out->select_rows = R->cap[2];
    out->n_advice_cells = R->n_advice_cells;
    out->n_permutations = R->perm_flat.size() / 2;
    out->n_dict = R->dict.size();
    out->n_fixed_patches = R->patch_flat.size() / 4;
    out->n_segments = 0;
    out->n_ops = R->n_ops;
    if (R->emit_shape) {
        h2e_records* W = const_cast<h2e_records*>(R);
        for (int reg = 0; reg < 3; reg++) {   // the views cover the arrays' whole capacity
            W->fix[reg].resize(R->cap[reg] * FIXC[reg], 0);
            W->flags[reg].resize(R->cap[reg] * ADVC[reg], 0);
        }
        out->dict = (const uint64_t*)R->dict.data();
        out->base_fix = R->fix[0].data();
        out->range_fix = R->fix[1].data();
        out->select_fix = R->fix[2].data();
        out->base_flags = R->flags[0].data();
        out->range_flags = R->flags[1].data();
        out->select_flags = R->flags[2].data();
        out->permutations = R->perm_flat.data();
        out->fixed_patches = R->patch_flat.data();
    }
    return 0;
}

// ---- ops: same names and argument meaning as the reference's traits -------------------------------------------------
int h2e_op_assign_w(h2e_records* R, const void* d_inputs, h2e_int* out, void* stream) {   // IntegerChipOps::assign_w (integer_chip.rs:236-258)
    if (!out) return fail(H2E_ERR_INVALID, "out is null");
    return records_op(R, key_of("assign_w", {}), 1, d_inputs, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t s) { *out = from_int(r.assign_w(s)); });
}
int h2e_op_assign(h2e_records* R, const void* d_inputs, uint32_t* out_cell, void* stream) {   // BaseChipOps::assign (base_chip.rs:351-355)
    if (!out_cell) return fail(H2E_ERR_INVALID, "out is null");
    return records_op(R, key_of("assign", {}), 1, d_inputs, stream, {OpOut{out_cell, sizeof(*out_cell)}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t s) { *out_cell = r.assign(s).ref; });
}
int h2e_op_int(h2e_records* R, int which, const h2e_int* a, const h2e_int* b, h2e_int* out, uint32_t* out_cond, void* stream) {
    const bool binary = which == H2E_INT_ADD || which == H2E_INT_SUB || which == H2E_INT_MUL || which == H2E_INT_DIV || which == H2E_INT_IS_EQUAL ||
                        which == H2E_INT_ASSERT_EQUAL;
    const bool no_out = which == H2E_INT_IS_ZERO || which == H2E_INT_IS_EQUAL || which == H2E_INT_ASSERT_EQUAL;
    if (!a || (binary && !b) || (!no_out && !out)) return fail(H2E_ERR_INVALID, "null operand");
    if (no_out) out = nullptr;
    return records_op(R, key_of("int", {{&which, sizeof(which)}, {a, sizeof(*a)}, {b, b ? sizeof(*b) : 0}}), 0, nullptr, stream, {OpOut{out, out ? sizeof(*out) : 0}, OpOut{out_cond, out_cond ? sizeof(*out_cond) : 0}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t) {
        h2e::AssignedInteger x = to_int(*a), y = b ? to_int(*b) : h2e::AssignedInteger();
        switch (which) {
            case H2E_INT_ADD: *out = from_int(r.int_add(x, y)); break;
            case H2E_INT_SUB: *out = from_int(r.int_sub(x, y)); break;
            case H2E_INT_MUL: *out = from_int(r.int_mul(x, y)); break;
            case H2E_INT_REDUCE: *out = from_int(r.reduce(x)); break;
            case H2E_INT_DIV: {
                auto d = r.int_div(x, y);
                *out = from_int(d.second);
                if (out_cond) *out_cond = d.first.v.ref;
            } break;
            case H2E_INT_NEG: *out = from_int(r.int_neg(x)); break;
            case H2E_INT_SQUARE: *out = from_int(r.int_square(x)); break;
            case H2E_INT_UNSAFE_INVERT: *out = from_int(r.int_unsafe_invert(x)); break;
            case H2E_INT_IS_ZERO: {
                h2e::AssignedCondition c = r.is_int_zero(x);
                if (out_cond) *out_cond = c.v.ref;
            } break;
            case H2E_INT_IS_EQUAL: {
                h2e::AssignedCondition c = r.is_int_equal(x, y);
                if (out_cond) *out_cond = c.v.ref;
            } break;
            case H2E_INT_ASSERT_EQUAL: r.assert_int_equal(x, y); break;
            default: throw std::runtime_error("h2e_op_int: unknown op");
        }
    });
}
int h2e_op_int_mul_small_constant(h2e_records* R, const h2e_int* a, uint64_t k, h2e_int* out, void* stream) {   // integer_chip.rs:618-658
    if (!a || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("int_mul_small", {{a, sizeof(*a)}, {&k, sizeof(k)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t) { *out = from_int(r.int_mul_small_constant(to_int(*a), k)); });
}
int h2e_op_assign_int_constant(h2e_records* R, const uint64_t* w_words, h2e_int* out, void* stream) {   // integer_chip.rs:580-598
    if (!w_words || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("int_const", {{w_words, (size_t)field_pair(R ? R->field_pair : 0).w_words * 8}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t) {
        *out = from_int(r.assign_int_constant(h2e::HBig::from_words(w_words, r.fp.w_words)));
    });
}
int h2e_op_bisec_int(h2e_records* R, uint32_t cond_cell, const h2e_int* a, const h2e_int* b, h2e_int* out, void* stream) {   // integer_chip.rs:660-681
    if (!a || !b || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("bisec_int", {{&cond_cell, sizeof(cond_cell)}, {a, sizeof(*a)}, {b, sizeof(*b)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t) {
        *out = from_int(r.bisec_int(h2e::AssignedCondition{h2e::AssignedValue{cond_cell}}, to_int(*a), to_int(*b)));
    });
}
namespace {
static h2e::AssignedFq2 to_fq2(const h2e_int* a) { return h2e::AssignedFq2{to_int(a[0]), to_int(a[1])}; }
static h2e::AssignedFq6 to_fq6(const h2e_int* a) { return h2e::AssignedFq6{to_fq2(a), to_fq2(a + 2), to_fq2(a + 4)}; }
static h2e::AssignedFq12 to_fq12(const h2e_int* a) { return h2e::AssignedFq12{to_fq6(a), to_fq6(a + 6)}; }
void from_fq2(const h2e::AssignedFq2& x, h2e_int* o) {
    o[0] = from_int(x.c0);
    o[1] = from_int(x.c1);
}
void from_fq6(const h2e::AssignedFq6& x, h2e_int* o) {
    from_fq2(x.c0, o);
    from_fq2(x.c1, o + 2);
    from_fq2(x.c2, o + 4);
}
void from_fq12(const h2e::AssignedFq12& x, h2e_int* o) {
    from_fq6(x.c0, o);
    from_fq6(x.c1, o + 6);
}
static std::unique_ptr<h2e::PairingOps> tower_of(h2e::Recorder& r) {
    if (r.fp.id == H2E_FIELD_BN256_FQ) return std::unique_ptr<h2e::PairingOps>(new h2e::Bn256PairingOps(r));
    if (r.fp.id == H2E_FIELD_BLS12_381_FQ) return std::unique_ptr<h2e::PairingOps>(new h2e::Bls12381PairingOps(r));
    throw std::runtime_error("no extension tower over this field");
}
}  // namespace
// Fq2ChipOps / Fq6ChipOps / Fq12ChipOps (src/circuit/fq12.rs:24-459) on assigned elements
int h2e_op_fq(h2e_records* R, int degree, int which, const h2e_int* a, const h2e_int* b, uint64_t imm, h2e_int* out, void* stream) {
    if (!a || (degree != 2 && degree != 6 && degree != 12)) return fail(H2E_ERR_INVALID, "bad argument");
    bool binary = which == H2E_FQ_ADD || which == H2E_FQ_SUB || which == H2E_FQ_MUL || which == H2E_FQ_ASSERT_EQUAL;
    if ((binary && !b) || (which != H2E_FQ_ASSERT_EQUAL && !out)) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("fq", {{&degree, sizeof(degree)}, {&which, sizeof(which)}, {a, sizeof(*a) * (size_t)degree}, {b, b ? sizeof(*b) * (size_t)degree : 0}, {&imm, sizeof(imm)}}), 0, nullptr, stream, {OpOut{out, out ? sizeof(*out) * (size_t)degree : 0}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t) {
        std::unique_ptr<h2e::PairingOps> t = tower_of(r);
        r.auto_cut_every = pairing_cut_every(r.fp.id == H2E_FIELD_BN256_FQ ? 0 : 1);
        auto bad = [] { throw std::runtime_error("h2e_op_fq: no such op at this degree"); };
        if (degree == 2) {
            h2e::AssignedFq2 x = to_fq2(a), y = b ? to_fq2(b) : h2e::AssignedFq2(), o;
            switch (which) {
                case H2E_FQ_ADD: o = t->fq2_add(x, y); break;
                case H2E_FQ_SUB: o = t->fq2_sub(x, y); break;
                case H2E_FQ_MUL: o = t->fq2_mul(x, y); break;
                case H2E_FQ_SQUARE: o = t->fq2_square(x); break;
                case H2E_FQ_NEG: o = t->fq2_neg(x); break;
                case H2E_FQ_DOUBLE: o = t->fq2_double(x); break;
                case H2E_FQ_CONJUGATE: o = t->fq2_conjugate(x); break;
                case H2E_FQ_UNSAFE_INVERT: o = t->fq2_unsafe_invert(x); break;
                case H2E_FQ_MUL_BY_NONRESIDUE: o = t->fq2_mul_by_nonresidue(x); break;
                case H2E_FQ_FROBENIUS_MAP: o = t->fq2_frobenius_map(x, (size_t)imm); break;
                case H2E_FQ_REDUCE: o = t->fq2_reduce(x); break;
                case H2E_FQ_ASSERT_EQUAL: t->fq2_assert_equal(x, y); return;
                default: bad();
            }
            from_fq2(o, out);
        } else if (degree == 6) {
            h2e::AssignedFq6 x = to_fq6(a), y = b ? to_fq6(b) : h2e::AssignedFq6(), o;
            switch (which) {
                case H2E_FQ_ADD: o = t->fq6_add(x, y); break;
                case H2E_FQ_SUB: o = t->fq6_sub(x, y); break;
                case H2E_FQ_MUL: o = t->fq6_mul(x, y); break;
                case H2E_FQ_SQUARE: o = t->fq6_square(x); break;
                case H2E_FQ_NEG: o = t->fq6_neg(x); break;
                case H2E_FQ_DOUBLE: o = t->fq6_double(x); break;
                case H2E_FQ_UNSAFE_INVERT: o = t->fq6_unsafe_invert(x); break;
                case H2E_FQ_MUL_BY_NONRESIDUE: o = t->fq6_mul_by_nonresidue(x); break;
                case H2E_FQ_FROBENIUS_MAP: o = t->fq6_frobenius_map(x, (size_t)imm); break;
                case H2E_FQ_REDUCE: o = t->fq6_reduce(x); break;
                case H2E_FQ_ASSERT_EQUAL: t->fq6_assert_equal(x, y); return;
                default: bad();
            }
            from_fq6(o, out);
        } else {
            h2e::AssignedFq12 x = to_fq12(a), y = b ? to_fq12(b) : h2e::AssignedFq12(), o;
            switch (which) {
                case H2E_FQ_ADD: o = t->fq12_add(x, y); break;
                case H2E_FQ_SUB: o = t->fq12_sub(x, y); break;
                case H2E_FQ_MUL: o = t->fq12_mul(x, y); break;
                case H2E_FQ_SQUARE: o = t->fq12_square(x); break;
                case H2E_FQ_NEG: o = t->fq12_neg(x); break;
                case H2E_FQ_DOUBLE: o = t->fq12_double(x); break;
                case H2E_FQ_CONJUGATE: o = t->fq12_conjugate(x); break;
                case H2E_FQ_UNSAFE_INVERT: o = t->fq12_unsafe_invert(x); break;
                case H2E_FQ_FROBENIUS_MAP: o = t->fq12_frobenius_map(x, (size_t)imm); break;
                case H2E_FQ_CYCLOTOMIC_SQUARE: o = t->fq12_cyclotomic_square(x); break;
                case H2E_FQ_REDUCE: o = t->fq12_reduce(x); break;
                case H2E_FQ_ASSERT_EQUAL: t->fq12_assert_eq(x, y); return;
                default: bad();
            }
            from_fq12(o, out);
        }
    });
}
int h2e_op_assign_points(h2e_records* R, uint32_t n, const void* d_inputs, h2e_point* out, void* stream) {   // EccChipBaseOps::assign_point x n
    if (!out || n == 0) return fail(H2E_ERR_INVALID, "bad argument");
    return records_op(R, key_of("assign_points", {{&n, sizeof(n)}}), 3 * n, d_inputs, stream, {OpOut{out, sizeof(*out) * (size_t)n}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& ecc, uint32_t s) {
        std::vector<h2e::AssignedPoint> pts = ecc.assign_points_from_inputs(n, s);
        for (uint32_t k = 0; k < n; k++) out[k] = from_point(pts[k]);
    });
}
int h2e_op_assign_scalars(h2e_records* R, uint32_t n, const void* d_inputs, h2e_int* out, void* stream) {   // ctx.assign / scalar_integer_ctx.assign_w x n
    if (!out || n == 0) return fail(H2E_ERR_INVALID, "bad argument");
    return records_op(R, key_of("assign_scalars", {{&n, sizeof(n)}}), n, d_inputs, stream, {OpOut{out, sizeof(*out) * (size_t)n}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& ecc, uint32_t s) {
        std::vector<h2e::AssignedInteger> sc = ecc.assign_scalars_from_inputs(n, s);
        for (uint32_t k = 0; k < n; k++) out[k] = from_int(sc[k]);
    });
}
int h2e_op_msm_unsafe(h2e_records* R, uint32_t n, const h2e_point* points, const h2e_int* scalars, const void* d_inputs, h2e_point* out,
                      void* stream) {   // EccChipScalarOps::msm_unsafe (ecc_chip.rs:373-408); inputs: generator (x, y), r1 (x, y), r2 (x, y)
    if (!points || !scalars || !out || n == 0) return fail(H2E_ERR_INVALID, "bad argument");
    return records_op(R, key_of("msm_unsafe", {{&n, sizeof(n)}, {points, sizeof(*points) * (size_t)n}, {scalars, sizeof(*scalars) * (size_t)n}}), 6, d_inputs, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& ecc, uint32_t s) {
        std::vector<h2e::AssignedPoint> pts;
        std::vector<h2e::AssignedInteger> sc;
        for (uint32_t k = 0; k < n; k++) {
            pts.push_back(to_point(points[k]));
            sc.push_back(to_int(scalars[k]));
        }
        h2e::NativeScalarEccContext::MsmInputs mi{s + 2, s + 3, s + 4, s + 5};
        *out = from_point(ecc.msm_unsafe(pts, sc, mi, s, s + 1));
    });
}
int h2e_op_ecc_assert_equal(h2e_records* R, const h2e_point* a, const h2e_point* b, void* stream) {   // ecc_chip.rs:644-658
    if (!a || !b) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("ecc_assert_equal", {{a, sizeof(*a)}, {b, sizeof(*b)}}), 0, nullptr, stream, {}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& ecc, uint32_t) { ecc.ecc_assert_equal(to_point(*a), to_point(*b)); });
}
// ---- the complete-addition / curvature surface of EccChipBaseOps ----
namespace {
static h2e::AssignedPointWithCurvature to_pc(const h2e_point_c& a) {
    return h2e::AssignedPointWithCurvature{to_int(a.p.x), to_int(a.p.y), h2e::AssignedCondition{h2e::AssignedValue{a.p.z}},
                                           h2e::AssignedCurvature{to_int(a.cv), h2e::AssignedCondition{h2e::AssignedValue{a.cz}}}};
}
h2e_point_c from_pc(const h2e::AssignedPointWithCurvature& a) {
    h2e_point_c r;
    r.p = from_point(a.to_point());
    r.cv = from_int(a.curvature.v);
    r.cz = a.curvature.z.v.ref;
    return r;
}
}  // namespace
int h2e_op_to_point_with_curvature(h2e_records* R, const h2e_point* a, h2e_point_c* out, void* stream) {   // ecc_chip.rs:695-708
    if (!a || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("to_point_with_curvature", {{a, sizeof(*a)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) { *out = from_pc(e.to_point_with_curvature(to_point(*a))); });
}
int h2e_op_ecc_reduce_with_curvature(h2e_records* R, const h2e_point* a, h2e_point_c* out, void* stream) {   // :677-693 (ecc_reduce :668-675, assign_identity :514-529)
    if (!a || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("ecc_reduce_with_curvature", {{a, sizeof(*a)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) { *out = from_pc(e.ecc_reduce_with_curvature(to_point(*a))); });
}
int h2e_op_ecc_double(h2e_records* R, const h2e_point_c* a, h2e_point* out, void* stream) {   // :630-642
    if (!a || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("ecc_double", {{a, sizeof(*a)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) { *out = from_point(e.ecc_double(to_pc(*a))); });
}
int h2e_op_ecc_add(h2e_records* R, const h2e_point_c* a, const h2e_point* b, h2e_point* out, void* stream) {   // :606-628
    if (!a || !b || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("ecc_add", {{a, sizeof(*a)}, {b, sizeof(*b)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) { *out = from_point(e.ecc_add(to_pc(*a), to_point(*b))); });
}
int h2e_op_ecc_neg(h2e_records* R, const h2e_point* a, h2e_point* out, void* stream) {   // :660-666
    if (!a || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("ecc_neg", {{a, sizeof(*a)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) { *out = from_point(e.ecc_neg(to_point(*a))); });
}
int h2e_op_ecc_encode(h2e_records* R, const h2e_point* a, uint32_t* out_cells3, void* stream) {   // :710-732
    if (!a || !out_cells3) return fail(H2E_ERR_INVALID, "null operand");
    if (R && field_pair(R->field_pair).limbs != 3) return fail(H2E_ERR_INVALID, "ecc_encode packs two 3-limb coordinates");
    return records_op(R, key_of("ecc_encode", {{a, sizeof(*a)}}), 0, nullptr, stream, {OpOut{out_cells3, 3 * sizeof(uint32_t)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) {
        std::vector<h2e::AssignedValue> v = e.ecc_encode(to_point(*a));
        for (int i = 0; i < 3; i++) out_cells3[i] = v[i].ref;
    });
}
int h2e_op_ecc_mul(h2e_records* R, const h2e_point* a, const h2e_int* scalar, const void* d_inputs, h2e_point* out, void* stream) {   // :418-420
    return h2e_op_msm_unsafe(R, 1, a, scalar, d_inputs, out, stream);
}
int h2e_op_assign_constant_point(h2e_records* R, const uint64_t* x_words, const uint64_t* y_words, int is_identity, h2e_point* out, void* stream) {   // :441-456
    if (!out || (!is_identity && (!x_words || !y_words))) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("assign_constant_point", {{x_words, x_words ? (size_t)field_pair(R ? R->field_pair : 0).w_words * 8 : 0}, {y_words, y_words ? (size_t)field_pair(R ? R->field_pair : 0).w_words * 8 : 0}, {&is_identity, sizeof(is_identity)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext& e, uint32_t) {
        h2e::HBig x, y;
        if (!is_identity) {
            x = h2e::HBig::from_words(x_words, r.fp.w_words);
            y = h2e::HBig::from_words(y_words, r.fp.w_words);
        }
        *out = from_point(e.assign_constant_point(x, y, is_identity != 0));
    });
}
int h2e_op_bisec_point_with_curvature(h2e_records* R, uint32_t cond_cell, const h2e_point_c* a, const h2e_point_c* b, h2e_point_c* out, void* stream) {   // :562-578
    if (!a || !b || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("bisec_point_with_curvature", {{&cond_cell, sizeof(cond_cell)}, {a, sizeof(*a)}, {b, sizeof(*b)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) {
        *out = from_pc(e.bisec_point_with_curvature(h2e::AssignedCondition{h2e::AssignedValue{cond_cell}}, to_pc(*a), to_pc(*b)));
    });
}
int h2e_op_assign_cache_point(h2e_records* R, const h2e_point_c* p, uint64_t group, uint64_t selector, void* stream) {   // :779-788
    if (!p) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("assign_cache_point", {{p, sizeof(*p)}, {&group, sizeof(group)}, {&selector, sizeof(selector)}}), 0, nullptr, stream, {}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) { e.assign_cache_point(to_pc(*p), (size_t)group, (size_t)selector); });
}
int h2e_op_assign_selected_point(h2e_records* R, uint32_t n, const h2e_point_c* candidates, uint32_t index_cell, uint64_t group, h2e_point_c* out,
                                 void* stream) {   // :790-812, the candidate picked on the device by the value of the index cell
    if (!candidates || !out || n == 0) return fail(H2E_ERR_INVALID, "bad argument");
    return records_op(R, key_of("assign_selected_point", {{&n, sizeof(n)}, {candidates, sizeof(*candidates) * (size_t)n}, {&index_cell, sizeof(index_cell)}, {&group, sizeof(group)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) {
        std::vector<h2e::AssignedPointWithCurvature> c;
        for (uint32_t k = 0; k < n; k++) c.push_back(to_pc(candidates[k]));
        *out = from_pc(e.assign_selected_point(c, h2e::AssignedValue{index_cell}, (size_t)group));
    });
}
int h2e_op_assign_g2_constant(h2e_records* R, const void* d_inputs, h2e_g2* out, void* stream) {   // fq2_assign_constant x 2 + assign_constant(0)
    if (!out) return fail(H2E_ERR_INVALID, "out is null");
    return records_op(R, key_of("assign_g2_constant", {}), 4, d_inputs, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t s) {
        out->x0 = from_int(r.assign_int_constant_input(s + 0));
        out->x1 = from_int(r.assign_int_constant_input(s + 1));
        out->y0 = from_int(r.assign_int_constant_input(s + 2));
        out->y1 = from_int(r.assign_int_constant_input(s + 3));
        out->z = r.assign_constant_u64(0).ref;
    });
}
int h2e_op_check_pairing(h2e_records* R, uint32_t n_pairs, const h2e_point* g1, const h2e_g2* g2, void* stream) {   // pairing_chip.rs:173-176
    if (!g1 || !g2 || n_pairs == 0) return fail(H2E_ERR_INVALID, "bad argument");
    if (R && R->field_pair == H2E_FIELD_BLS12_381_FR) return fail(H2E_ERR_INVALID, "no pairing over this field");
    return records_op(R, key_of("check_pairing", {{&n_pairs, sizeof(n_pairs)}, {g1, sizeof(*g1) * (size_t)n_pairs}, {g2, sizeof(*g2) * (size_t)n_pairs}}), 0, nullptr, stream, {}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t) {
        r.auto_cut_every = pairing_cut_every(r.fp.id == H2E_FIELD_BN256_FQ ? 0 : 1);
        std::unique_ptr<h2e::PairingOps> po;
        if (r.fp.id == H2E_FIELD_BN256_FQ) po.reset(new h2e::Bn256PairingOps(r));
        else po.reset(new h2e::Bls12381PairingOps(r));
        std::vector<h2e::AssignedPoint> a;
        std::vector<h2e::AssignedG2Affine> b;
        for (uint32_t k = 0; k < n_pairs; k++) {
            a.push_back(to_point(g1[k]));
            b.push_back(h2e::AssignedG2Affine{h2e::AssignedFq2{to_int(g2[k].x0), to_int(g2[k].x1)}, h2e::AssignedFq2{to_int(g2[k].y0), to_int(g2[k].y1)},
                                              h2e::AssignedCondition{h2e::AssignedValue{g2[k].z}}});
        }
        std::vector<h2e::PairingOps::Term> terms;
        for (uint32_t k = 0; k < n_pairs; k++) terms.push_back(h2e::PairingOps::Term(&a[k], &b[k]));
        po->check_pairing(terms);
    });
}
int h2e_op_pairing(h2e_records* R, uint32_t n_pairs, const h2e_point* g1, const h2e_g2* g2, h2e_int* out12, void* stream) {   // pairing_chip.rs:157-171
    if (!g1 || !g2 || !out12 || n_pairs == 0) return fail(H2E_ERR_INVALID, "bad argument");
    if (R && R->field_pair == H2E_FIELD_BLS12_381_FR) return fail(H2E_ERR_INVALID, "no pairing over this field");
    return records_op(R, key_of("pairing", {{&n_pairs, sizeof(n_pairs)}, {g1, sizeof(*g1) * (size_t)n_pairs}, {g2, sizeof(*g2) * (size_t)n_pairs}}), 0, nullptr, stream, {OpOut{out12, sizeof(*out12) * 12}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t) {
        r.auto_cut_every = pairing_cut_every(r.fp.id == H2E_FIELD_BN256_FQ ? 0 : 1);
        std::unique_ptr<h2e::PairingOps> po = tower_of(r);
        std::vector<h2e::AssignedPoint> a;
        std::vector<h2e::AssignedG2Affine> b;
        for (uint32_t k = 0; k < n_pairs; k++) {
            a.push_back(to_point(g1[k]));
            b.push_back(h2e::AssignedG2Affine{h2e::AssignedFq2{to_int(g2[k].x0), to_int(g2[k].x1)}, h2e::AssignedFq2{to_int(g2[k].y0), to_int(g2[k].y1)},
                                              h2e::AssignedCondition{h2e::AssignedValue{g2[k].z}}});
        }
        std::vector<h2e::PairingOps::Term> terms;
        for (uint32_t k = 0; k < n_pairs; k++) terms.push_back(h2e::PairingOps::Term(&a[k], &b[k]));
        from_fq12(po->pairing(terms), out12);
    });
}

