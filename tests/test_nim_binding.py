"""CPU suite, part 4: the Nim side of the boundary, checked mechanically (no Nim compiler exists in the build image, so
nim/codex_p2.nim and nim/overlay/src/** have never been compiled):

  * every `{.importc.}` proc of nim/codex_p2.nim against its prototype in include/codex_p2.h: name, arity, return type and,
    argument by argument, width and pointer-ness (csize_t <-> size_t, uint64 <-> uint64_t, ptr byte <-> uint8_t*, cint <-> int,
    the opaque handle KINDS, Cp2Config's field order and types against cp2_config);
  * every call of a cp2_* function in the Nim sources passes as many arguments as the prototype takes;
  * every overlay module exports exactly the public procs / types of the reference module it replaces, with the same
    signatures (the live tree under /root/reference when present, else the committed interface list);
  * no `unsafeAddr x[0]` on a possibly empty input is left in the binding.
"""
import json
import os
import re

import pytest

import nim_api as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NIM = os.path.join(ROOT, "codex-storage-proofs-circuits_amd", "nim")
HEADER = open(os.path.join(ROOT, "include", "codex_p2.h")).read()
BINDING = open(os.path.join(NIM, "codex_p2.nim")).read()


def test_every_importc_matches_its_header_prototype():
    protos, procs = N.header_prototypes(HEADER), N.nim_importc(BINDING)
    assert len(procs) >= 30
    for name, (ret, args) in procs.items():
        assert name in protos, "nim/codex_p2.nim imports %s, which include/codex_p2.h does not declare" % name
        cret, cargs = protos[name]
        assert "?" not in ret and all("?" not in a for a in args), (name, ret, args)
        assert len(args) == len(cargs), "%s: %d parameters in Nim, %d in the header" % (name, len(args), len(cargs))
        assert ret == cret, "%s: returns %s in Nim, %s in the header" % (name, ret, cret)
        for i, (a, c) in enumerate(zip(args, cargs)):
            assert a == c, "%s: parameter %d is %s in Nim, %s in the header" % (name, i, a, c)
    # the header itself parsed completely: no unknown type anywhere
    for name, (ret, args) in protos.items():
        assert "?" not in ret and all("?" not in a for a in args), (name, ret, args)


def test_the_parsers_see_drift():
    """The check above is only worth something if a wrong width, a missing parameter or a swapped handle fails it."""
    protos = N.header_prototypes(HEADER)
    bad = {
        "width": BINDING.replace("proc cp2_felts_per_bytes(len: csize_t): csize_t", "proc cp2_felts_per_bytes(len: uint32): csize_t"),
        "arity": BINDING.replace("proc cp2_merkle_root(ctx: Cp2Ctx, leaves: ptr byte, n: csize_t, outp: ptr byte)",
                                 "proc cp2_merkle_root(ctx: Cp2Ctx, leaves: ptr byte, outp: ptr byte)"),
        "handle": BINDING.replace("proc cp2_dataset_free(ds: Cp2Dataset)", "proc cp2_dataset_free(ds: Cp2ProofInput)"),
        "pointer": BINDING.replace("proc cp2_cell_indices(ctx: Cp2Ctx, entropy, slotRoot: ptr byte, nCells: uint64",
                                   "proc cp2_cell_indices(ctx: Cp2Ctx, entropy, slotRoot: ptr byte, nCells: ptr uint64"),
    }
    for what, text in bad.items():
        assert text != BINDING, what
        procs = N.nim_importc(text)
        assert any(protos[k] != v for k, v in procs.items()), what


def test_config_struct_layout_matches():
    c, n = N.header_config_fields(HEADER), N.nim_config_fields(BINDING)
    snake = lambda s: re.sub(r"(?<!^)([A-Z])", r"_\1", s).lower().replace("log2_n_slots", "log2_nslots")   # noqa: E731
    assert [(snake(name), t) for name, t in n] == c
    # and the ctypes mirror the tests use
    import __graft_entry__ as g
    import ctypes
    pkg = g.load_package()
    widths = {"i32": ctypes.c_int32, "u64": ctypes.c_uint64, "cstr": ctypes.c_char_p}
    assert [(name, widths[t]) for name, t in c] == list(pkg.Config._fields_)


def test_every_call_site_passes_the_prototypes_arity():
    protos = N.header_prototypes(HEADER)
    calls = N.nim_call_arities(BINDING)
    assert len(calls) >= 25
    for name, n_args in calls:
        assert name in protos and n_args == len(protos[name][1]), (name, n_args, len(protos[name][1]))


def _reference_api():
    live = "/root/reference/reference/nim/proof_input/src"
    fixture = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_nim_api.json")))["modules"]
    if os.path.isdir(live):
        api = {m: [list(t) for t in N.public_api(open(os.path.join(live, m)).read())] for m in fixture}
        assert api == fixture, "tests/golden/reference_nim_api.json is stale: run tests/golden/make_reference_api.py"
    return fixture


@pytest.mark.parametrize("module", ["types/bn254.nim", "merkle/bn254.nim", "blocks/bn254.nim", "sample/bn254.nim",
                                    "gen_input/bn254.nim", "json/bn254.nim"])
def test_overlay_module_exports_what_the_reference_module_exports(module):
    want = [tuple(t) for t in _reference_api()[module]]
    got = N.public_api(open(os.path.join(NIM, "overlay", "src", module)).read())
    assert [(k, n) for k, n, _ in got] == [(k, n) for k, n, _ in want], module          # the same names, nothing more, nothing less
    for (k, name, sig), (_, _, ref_sig) in zip(got, want):
        assert sig == ref_sig, "%s: %s is %s in the overlay, %s in the reference" % (module, name, sig, ref_sig)


def test_no_index_into_a_possibly_empty_input():
    """`unsafeAddr x[0]` raises IndexDefect on an empty openArray where nim-poseidon2 hashes the padding (and the C ABI
    accepts length 0): all such addresses go through the `firstByte` template."""
    code = N.strip_nim_comments(BINDING)
    uses = re.findall(r"unsafeAddr\s+\w+\[0\]", code)
    assert len(uses) == 1 and "template firstByte" in code        # the one inside the template, guarded by a.len > 0
    for m in ("merkle/bn254.nim", "blocks/bn254.nim", "sample/bn254.nim", "gen_input/bn254.nim", "json/bn254.nim", "types/bn254.nim"):
        assert "unsafeAddr" not in N.strip_nim_comments(open(os.path.join(NIM, "overlay", "src", m)).read()), m


def test_cli_imports_are_what_integration_md_says():
    """cli.nim also imports the Goldilocks modules (and through them nim-goldilocks-hash): INTEGRATION.md must say that the
    overlay leaves those imports -- and their nimble pins -- in place."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "goldilocks" in text.lower() and "nimble" in text.lower()
