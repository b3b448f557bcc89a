"""Host-side replay of the packed expansion's scheduling (engine.hip h2e_run_tape_packed): for the tape of a pairing check, how many
steps does a wave of G sub-ranges take under a given pick rule, and what do they cost in 'cells written' units, against G separate waves?"""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
from halo2ecc_s_amd import Program
from halo2ecc_s_amd.engine import lib

NAMES = ["NOP", "ASSIGN_W", "ASSIGN", "ASSIGN_BIT", "CONST_INT", "CONST_INT_INPUT", "CONST", "INT_ADD", "INT_SUB", "INT_NEG", "INT_MUL_SMALL", "INT_MUL", "REDUCE",
         "IS_INT_ZERO", "NOT", "MASK_INT", "DIV_CORE", "BISEC_INT", "SUM_LIMBS", "ASSERT_CONST", "BISEC", "AND", "OR", "XNOR", "DECOMPOSE_NATIVE", "PICK_INDEX",
         "CACHE_INT", "SELECT_POINT", "DECOMPOSE_LIMB", "SHIFT_ADD"]
RANK = [0, 20, 1, 2, 12, 13, 3, 14, 15, 16, 17, 30, 28, 22, 4, 18, 31, 19, 5, 6, 7, 8, 9, 10, 26, 11, 21, 23, 25, 24]
# advice cells an op writes (bn256) ~ its instruction count
CELLS = {"INT_MUL": 125, "DIV_CORE": 140, "REDUCE": 40, "INT_ADD": 13, "INT_SUB": 13, "INT_NEG": 10, "INT_MUL_SMALL": 10, "IS_INT_ZERO": 40, "ASSIGN_W": 23,
         "MASK_INT": 12, "BISEC_INT": 20, "CONST_INT": 4, "CONST_INT_INPUT": 4}


def tape(prog, launch):
    L = lib()
    L.h2e_program_tape_opcodes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    n = L.h2e_program_tape_opcodes(prog._h, launch, None, 0, None, 0, None)
    ops = np.zeros(n, dtype=np.uint16)
    subs = np.zeros(n + 2, dtype=np.uint32)
    ns = C.c_uint32(0)
    L.h2e_program_tape_opcodes(prog._h, launch, ops.ctypes.data, n, subs.ctypes.data, n + 2, C.byref(ns))
    return ops, subs[:ns.value]


def cost(opc):
    return CELLS.get(NAMES[opc], 3)


def simulate(ops, subs, G, rank=RANK, order=None):
    nsub = len(subs) - 1
    order = list(range(nsub)) if order is None else order
    tot_steps = tot_cost = tot_lane_cost = 0
    for w0 in range(0, nsub, G):
        grp = order[w0:w0 + G]
        pos = [int(subs[s]) for s in grp]
        end = [int(subs[s + 1]) for s in grp]
        while True:
            heads = [rank[ops[p]] if p < e else 255 for p, e in zip(pos, end)]
            pick = min(heads)
            if pick == 255:
                break
            k = heads.index(pick)
            c = cost(ops[pos[k]])
            n_exec = 0
            for i in range(len(grp)):
                if heads[i] == pick:
                    pos[i] += 1
                    n_exec += 1
            tot_steps += 1
            tot_cost += c
            tot_lane_cost += c * n_exec
    return tot_steps, tot_cost, tot_lane_cost


def recut(n_ops, every):
    return np.array(list(range(0, n_ops, every)) + [n_ops], dtype=np.uint32)


def wave_costs(ops, subs, G):
    nsub = len(subs) - 1
    out = []
    for w0 in range(0, nsub, G):
        grp = list(range(w0, min(nsub, w0 + G)))
        pos = [int(subs[s]) for s in grp]
        end = [int(subs[s + 1]) for s in grp]
        tot = 0
        while True:
            heads = [RANK[ops[p]] if p < e else 255 for p, e in zip(pos, end)]
            pick = min(heads)
            if pick == 255:
                break
            tot += cost(ops[pos[heads.index(pick)]])
            for i in range(len(grp)):
                if heads[i] == pick:
                    pos[i] += 1
        out.append(tot)
    return np.array(out)


def makespan(costs, slots):
    """list scheduling: every SIMD slot takes the next wave when it is free"""
    import heapq
    h = [0] * slots
    heapq.heapify(h)
    for c in costs:
        heapq.heappush(h, heapq.heappop(h) + int(c))
    return max(h)


def orders(curve, G, slots):
    """what the order of the sub-ranges is worth (h2e_capi.cpp pack_orders_of): tape order (adjacent sub-ranges share a wave) against
    the engine's order tables (waves of one opcode sequence, heaviest first), as the work of the longest of `slots` SIMDs"""
    prog = Program.pairing_check_bn256(emit_shape=False) if curve == "bn256" else Program.pairing_check_bls12_381(emit_shape=False)
    for li, l in enumerate(prog.launches()):
        if l["n_ops"] < 1000:
            continue
        ops, subs = tape(prog, li)
        nsub = len(subs) - 1
        seq = {ops[subs[k]:subs[k + 1]].tobytes() for k in range(nsub)}
        print(f"{curve} launch {li}: {len(ops)} ops, {nsub} sub-ranges, {len(seq)} different opcode sequences, G = {G}, {slots} SIMD slots")
        wc = wave_costs(ops, subs, G)
        print(f"  tape order:   {len(wc)} waves, work {wc.sum()}, longest wave {wc.max()}, longest SIMD {makespan(wc, slots)} (even split: {wc.sum() / slots:.0f})")
        tab = prog.pack_order(li, G)
        wc2 = []
        for w in tab:
            first = int(w[0])
            wc2.append(sum(cost(o) for o in ops[subs[first]:subs[first + 1]]))   # one sequence: the wave costs what one sub-range costs
        wc2 = np.array(wc2)
        print(f"  order tables: {len(wc2)} waves, work {wc2.sum()}, longest wave {wc2.max()}, longest SIMD {makespan(wc2, slots)} (even split: {wc2.sum() / slots:.0f})")


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "orders":
    orders(sys.argv[1], int(sys.argv[3]), int(sys.argv[4]))
    sys.exit(0)

if __name__ == "__main__":
    curve = sys.argv[1] if len(sys.argv) > 1 else "bn256"
    prog = Program.pairing_check_bn256(emit_shape=False) if curve == "bn256" else Program.pairing_check_bls12_381(emit_shape=False)
    launches = prog.launches()
    dom = max(range(len(launches)), key=lambda i: launches[i]["cells"])
    ops, subs = tape(prog, dom)
    print(curve, "launch", dom, "ops", len(ops), "sub-ranges", len(subs) - 1)
    hist = np.bincount(ops, minlength=30)
    print({NAMES[i]: int(hist[i]) for i in range(30) if hist[i]})
    base_steps, base_cost, _ = simulate(ops, subs, 1)
    print("G = 1: steps", base_steps, "cost", base_cost)
    for G in (2, 4, 8, 32):
        st, co, lc = simulate(ops, subs, G)
        print(f"G = {G}: steps {st} ({st / base_steps:.2f} of separate waves), cost {co} ({co / base_cost:.2f}), lane utilisation {lc / (co * G):.2f}")


