#!/usr/bin/env python3
"""Generates tests/golden/reference_nim_api.json: the PUBLIC interface (exported names and their normalised signatures, nothing
else) of the reference modules that nim/overlay/src/** replaces, read from /root/reference as text.  tests/test_nim_binding.py
compares the overlay against the live reference tree when it exists (this container) and against this list elsewhere.

  python tests/golden/make_reference_api.py [/root/reference]
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import nim_api  # noqa: E402

MODULES = ["types/bn254.nim", "merkle/bn254.nim", "blocks/bn254.nim", "sample/bn254.nim", "gen_input/bn254.nim", "json/bn254.nim"]


def reference_api(ref_root):
    src = os.path.join(ref_root, "reference", "nim", "proof_input", "src")
    return {m: [list(t) for t in nim_api.public_api(open(os.path.join(src, m)).read())] for m in MODULES}


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = {"note": "exported names + normalised signatures of reference/nim/proof_input/src/<module> (interface only)", "modules": reference_api(ref)}
    with open(os.path.join(HERE, "reference_nim_api.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    for m, api in out["modules"].items():
        print(m, len(api))
