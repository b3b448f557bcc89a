"""The drop-in boundary's three texts must name the same functions: include/h2e.h (the C ABI), what libh2e.so exports, and the
Rust-side binding a maintainer of the reference would add (integration/rust/src/gpu/ffi.rs - source only: the image has no Rust
toolchain).  Round 5's review found 15 header functions missing from ffi.rs and 39 internal symbols exported by the library."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "h2e.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(h2e_[a-z0-9_]+)\s*\(", text))


def test_rust_ffi_declares_every_header_function_and_nothing_else():
    ffi = open(os.path.join(ROOT, "integration", "rust", "src", "gpu", "ffi.rs")).read()
    declared = set(re.findall(r"pub fn (h2e_[a-z0-9_]+)\s*\(", ffi))
    hdr = _header_functions()
    assert hdr - declared == set(), f"missing from ffi.rs: {sorted(hdr - declared)}"
    assert declared - hdr == set(), f"ffi.rs declares what the header does not: {sorted(declared - hdr)}"


def test_rust_ffi_argument_counts_match_the_header():
    """same number of parameters per function on both sides (types are checked by eye / by rustc the day there is one)"""
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "h2e.h")).read(), flags=re.S)
    ffi = re.sub(r"//[^\n]*", "", open(os.path.join(ROOT, "integration", "rust", "src", "gpu", "ffi.rs")).read())

    def n_args(body):
        body = body.strip()
        return 0 if body in ("", "void") else body.count(",") + 1
    c_sig = {m.group(1): n_args(m.group(2)) for m in re.finditer(r"\b(h2e_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text)}
    r_sig = {m.group(1): n_args(m.group(2)) for m in re.finditer(r"pub fn (h2e_[a-z0-9_]+)\s*\(([^()]*)\)", ffi)}
    assert len(c_sig) >= 70
    bad = {f: (c_sig[f], r_sig.get(f)) for f in c_sig if r_sig.get(f) != c_sig[f]}
    assert not bad, bad


def test_library_exports_exactly_the_header():
    """-fvisibility=hidden + H2E_API on the header's functions: the dynamic symbol table of libh2e.so holds the header's functions
    (+ the test hooks the header documents under "Test hooks"), not the engine's internal launchers"""
    lib = os.path.join(ROOT, "halo2ecc_s_amd", "libh2e.so")
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    h2e = {s for s in exported if s.startswith("h2e_")}
    hdr = _header_functions()
    assert hdr - h2e == set(), f"declared but not exported: {sorted(hdr - h2e)}"
    extra = h2e - hdr
    assert extra == set(), f"exported but not in include/h2e.h: {sorted(extra)}"
