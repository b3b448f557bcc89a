// struct h2e_program: a recorded program + what the compiler passes make of it (declarations; the passes are in program_value_chain.hpp,
// program_replay.hpp and program_schedule.hpp).
#pragma once
struct h2e_program {
    int field_pair;
    std::unique_ptr<h2e::Recorder> rec;
    uint64_t base_rows = 0, range_rows = 0, select_rows = 0;
    std::vector<uint32_t> perm_flat, patch_flat;
    // device copies (per device), created on first run
    int device = -1;
    H2EOp* d_tape = nullptr;
    uint32_t* d_aux = nullptr;
    uint64_t* d_pool = nullptr;
    uint32_t* d_params = nullptr;
    uint32_t* d_fixups = nullptr;
    uint32_t* d_pre_args = nullptr;
    uint32_t* d_subs = nullptr;
    std::vector<uint32_t> h_subs;          // per segment with cuts: [0, cut_1, ..., n_ops]
    std::vector<uint32_t> seg_sub_begin;   // per segment: index into h_subs (or ~0u)
    std::vector<uint32_t> seg_n_sub;
    // order tables of the packed expansion (tape.h H2ELaunch::pk_order), per cut segment and group count 2 << k
    std::vector<uint32_t> h_pk_order;
    std::vector<std::array<uint32_t, 5>> seg_pk_off, seg_pk_waves;
    uint32_t* d_pk_order = nullptr;
    bool pk_built = false;
    std::vector<uint8_t> seg_deferrable;   // a segment without cuts whose cells no later kernel reads: runs off the critical stream
    // compiled values-only replay (tape.h "V-tape"), per cut segment
    std::vector<H2EVRec> h_vtape;
    std::vector<uint32_t> seg_v_slots, seg_v_units;        // per segment: LDS sizing of the replay kernel
    std::vector<uint32_t> seg_piece_begin, seg_n_pieces;   // per segment: pieces = [first record, end record) pairs in h_vpieces
    std::vector<uint32_t> h_vpieces;
    H2EVRec* d_vtape = nullptr;
    uint32_t* d_vpieces = nullptr;
    // level-parallel replay (segments whose dependency graph is much shallower than it is long: the pairings)
    std::vector<H2EVRec> h_lrecs;               // 64 records per step (lane l of a step runs record 64 * step + l)
    std::vector<uint32_t> h_lrefs;              // cell refs of global integer operands
    std::vector<uint32_t> seg_l_begin, seg_l_steps, seg_l_slots, seg_l_pair;
    // hint store (field_chain.hpp): per segment with field hints, in place of a compiled replay
    std::vector<uint32_t> h_swords, h_soffsets, seg_s_begin, seg_so_begin, seg_sk_begin, seg_n_sops;
    std::vector<uint32_t> h_sext, seg_sx_begin;   // extension leaves of the store records (tape.h H2EStoreExt)
    uint32_t* d_sext = nullptr;
    std::vector<uint64_t> h_sktab;
    uint32_t *d_swords = nullptr, *d_soffsets = nullptr;
    uint64_t* d_sktab = nullptr;
    std::vector<uint32_t> h_lrounds;            // wave mode: per round (first record, count | kind << 8)
    std::vector<uint32_t> seg_lr_begin, seg_l_recs;
    H2EVRec* d_lrecs = nullptr;
    uint32_t* d_lrefs = nullptr;
    uint32_t* d_lrounds = nullptr;
    int64_t tail_from = -1;   // first segment of the program's serial tail (runs on the job slot's side stream), -1: none
    uint8_t* d_flags[3] = {nullptr, nullptr, nullptr};   // assigned / permute bytes on the device (h2e_export masks with them)
    // shape artefacts on the device (h2e_export_fixed / h2e_export_copy_constraints), uploaded on first use
    uint32_t* d_fix[3] = {nullptr, nullptr, nullptr};
    uint64_t* d_dict = nullptr;
    uint32_t* d_patches = nullptr;
    uint32_t* d_perms = nullptr;
    // h2e_check: base fixed ids with the cells made from instance inputs marked (bit 31 | patch index), the dictionary and the
    // range gates' shifts in Montgomery form, the select chip's table rows sorted by their encode cell, per-instance patch values
    uint32_t* d_fix_ck = nullptr;
    uint64_t *d_dict_m = nullptr, *d_shifts_m = nullptr, *d_sel_keys = nullptr, *d_patch_vals = nullptr;
    uint32_t* d_sel_key_rows = nullptr;
    uint32_t n_sel_keys = 0;
    size_t patch_vals_cap = 0;
    bool check_ready = false;

    ~h2e_program() {
        if (device >= 0) {
            (void)hipFree(d_tape);
            (void)hipFree(d_aux);
            (void)hipFree(d_pool);
            (void)hipFree(d_params);
            (void)hipFree(d_fixups);
            (void)hipFree(d_pre_args);
            (void)hipFree(d_subs);
            (void)hipFree(d_pk_order);
            (void)hipFree(d_vtape);
            (void)hipFree(d_vpieces);
            (void)hipFree(d_lrecs);
            (void)hipFree(d_lrefs);
            (void)hipFree(d_lrounds);
            (void)hipFree(d_swords);
            (void)hipFree(d_soffsets);
            (void)hipFree(d_sktab);
            (void)hipFree(d_sext);
            for (int i = 0; i < 3; i++) (void)hipFree(d_flags[i]);
        }
        for (int i = 0; i < 3; i++) (void)hipFree(d_fix[i]);
        (void)hipFree(d_dict);
        (void)hipFree(d_patches);
        (void)hipFree(d_perms);
        (void)hipFree(d_fix_ck);
        (void)hipFree(d_dict_m);
        (void)hipFree(d_shifts_m);
        (void)hipFree(d_sel_keys);
        (void)hipFree(d_patch_vals);
        (void)hipFree(d_sel_key_rows);
    }
    // Liveness over sub-ranges: an arithmetic op whose result cells are only read by ops of its own sub-range gets
    // H2E_FLAG_LOCAL_RESULT, so the values-only replay keeps that result in LDS and does not store it (the full
    // expansion of the sub-range recomputes and stores it anyway).  Row ownership: an op owns the rows from its
    // first row up to the next op's first row.  Every reference that can reach a cut segment is considered: op
    // refs of all segments, candidate tables (aux), strand parameters and the program's outputs.
    void mark_local_results();

    // Compile one cut segment into V-tape records (tape.h).  Values = results of alive ops; each gets an LDS slot for
    // as long as later ops of the replay read it (furthest-next-use eviction when the slots run out: an evicted or
    // never cached value goes through its cells, so its producer stores it).
    void compile_replay(const h2e::Segment* sg, const H2EOp* ops, uint32_t n_ops, const uint32_t* first, const uint32_t* last,
                        const std::function<int(uint32_t, uint32_t)>& producer);

    // A segment without cuts normally runs on the caller's (critical) stream because later value-chain kernels may
    // read any of its cells.  If no reference anywhere (later ops, strand parameters, candidate tables, predictor
    // arguments, outputs) points into its rows, it can run on the expansion stream instead.
    void mark_deferrable();

    // A fork segment without cuts that only reads segments without cuts (the scalar decomposition: it reads the
    // assigned scalars, and only the MSM windows read its bits) need not sit in the value chain between its neighbours:
    // it runs on a side stream as soon as the last segment it reads is done, and the first segment that reads it waits.
    std::vector<int32_t> seg_side_dep;      // -2: not a side segment; else index of the last segment it depends on (-1: none)
    std::vector<uint32_t> seg_first_reader; // for side segments: first later segment that references its rows
    void mark_side_segments();

    // Expansion result cache (engine.hip ld_int_x / xc_put_x): per sub-range of a cut segment, which of the three LDS
    // entries an integer result goes to and which operands are read from them - furthest-next-use replacement over the
    // static op sequence.  Encoded in op.flags bits 8-15.
    void assign_expansion_slots();

    void finish();
};

#include "program_value_chain.hpp"
#include "program_replay.hpp"
#include "program_schedule.hpp"
