"""Build the engine's shared library in-tree (hipcc, gfx950).  `python -m halo2ecc_s_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libh2e.so")
DEPS = ["tape.h", "wide_int.h", "modinv62.h", "hbig.hpp", "recorder.hpp", "recorder_ecc.hpp", "recorder_pairing.hpp",
        "pairing_constants.hpp", "field_chain.hpp", os.path.join("..", "..", "include", "h2e.h")]
# the C-ABI layer's translation unit in parts (h2e_capi.cpp includes them): only that unit depends on these
CAPI_DEPS = ["capi_common.hpp", "program.hpp", "program_value_chain.hpp", "program_replay.hpp", "program_schedule.hpp", "run_state.hpp", "run.hpp",
             "ring.hpp", "records_api.hpp"]


EXPORT_MAP = os.path.join(CSRC, "libh2e.map")


def write_export_map():
    """The library's ABI surface is include/h2e.h and nothing else: a linker version script with the header's function names as the
    only global symbols (the engine units' h2e_engine_* launchers stay internal to the shared object)."""
    import re
    with open(os.path.join(HERE, "..", "include", "h2e.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(h2e_[a-z0-9_]+)\s*\(", text)))
    body = "{\n  global:\n" + "".join(f"    {n};\n" for n in names) + "  local:\n    *;\n};\n"
    if not os.path.exists(EXPORT_MAP) or open(EXPORT_MAP).read() != body:
        with open(EXPORT_MAP, "w") as f:
            f.write(body)
    return EXPORT_MAP


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    """engine.hip is compiled once per field pair (-DH2E_FP_ONLY=k: ~2 minutes each, side by side) + the C-ABI layer"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    deps = [os.path.join(CSRC, d) for d in DEPS]
    objs = []
    running = []
    units = ([("engine.hip", f"engine_fp{k}.o", [f"-DH2E_FP_ONLY={k}"]) for k in range(3)] + [("h2e_capi.cpp", "h2e_capi.o", [])]
             + [("engine.hip", f"engine_cols_fp{k}.o", [f"-DH2E_FP_ONLY={k}", "-DH2E_COLS"]) for k in range(3)]   # the column-emission units
             + [("checker.hip", "checker.o", [])]   # the device-side constraint check: a unit of its own, no code shared with the engine
             + [("handoff.hip", "handoff.o", [])])  # field-independent hand-off kernels (unit records)
    for src, obj, defs in units:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, obj)
        if force or _stale(o, [s] + deps + ([os.path.join(CSRC, d) for d in CAPI_DEPS] if src == "h2e_capi.cpp" else [])):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + defs + ["-c", s, "-o", o]
            if src.endswith(".cpp"):
                cmd.insert(1, "-x")
                cmd.insert(2, "hip")
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((cmd, subprocess.Popen(cmd)))
        objs.append(o)
    for cmd, proc in running:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    emap = write_export_map()
    if force or _stale(LIB, objs + [emap]):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={emap}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
