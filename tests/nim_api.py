"""Text-level readers of Nim sources and of include/codex_p2.h for tests/test_nim_binding.py (no Nim compiler exists in the
build image, so the binding is checked mechanically: names, arity, and the width / pointer-ness of every argument)."""
import re


def strip_nim_comments(text):
    out = []
    for line in text.splitlines():
        # a '#' outside a string literal starts a comment (good enough for these sources: no '#' inside their literals
        # except within "..." which we skip)
        buf, in_str, i = [], False, 0
        while i < len(line):
            ch = line[i]
            if ch == '"':
                in_str = not in_str
            if ch == '#' and not in_str:
                break
            buf.append(ch)
            i += 1
        out.append("".join(buf))
    return "\n".join(out)


def public_api(text):
    """[(kind, name, normalised signature)] of every exported (`*`) proc / func / iterator / type of a Nim module.
    kind: 'routine' (proc and func are interchangeable for a caller) or 'type'.  The signature is everything from the
    parameter list to the return type, without whitespace and lower-cased (Nim identifiers are style-insensitive)."""
    text = strip_nim_comments(text)
    api = []
    for m in re.finditer(r"^(proc|func|iterator)\s+(`?[\w]+`?)\*\s*(\[[^\]]*\])?\s*\(", text, flags=re.M):
        # find the matching ')' of the parameter list
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        params = text[m.end():i - 1]
        rest = text[i:]
        ret = ""
        r = re.match(r"\s*:\s*([^=\n{]+?)\s*(=|\{\.|$)", rest, flags=re.M)
        if r:
            ret = r.group(1)
        norm = re.sub(r"\s+", "", "(%s):%s" % (params, ret)).lower()
        api.append(("iterator" if m.group(1) == "iterator" else "routine", m.group(2), norm))
    for m in re.finditer(r"^type\s+(\w+)\*\s*=\s*([^\n]+)$", text, flags=re.M):
        api.append(("type", m.group(1), re.sub(r"\s+", "", m.group(2)).lower()))
    return sorted(api)


# ---- C header ---------------------------------------------------------------------------------------------------
HANDLES = {"cp2_ctx": "ctx", "cp2_dataset": "dataset", "cp2_proof_input": "proof_input", "cp2_slot_trees": "slot_trees",
           "cp2_multi": "multi", "cp2_multi_dataset": "multi_dataset"}


def c_type_class(t, array=False):
    t = re.sub(r"\bconst\b", "", t).strip()
    t = re.sub(r"\s+", " ", t).replace(" *", "*")
    stars = t.count("*") + (1 if array else 0)
    base = t.replace("*", "").strip()
    scalar = {"int": "i32", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "size_t": "usize", "void": "void",
              "uint8_t": "u8", "char": "char"}
    if base in HANDLES:
        return ("handle:" + HANDLES[base],) * 1 if stars == 1 else "ptr(handle:%s)" % HANDLES[base] if stars == 2 else "?"
    if base == "cp2_config":
        return "ptr(config)" if stars == 1 else "?"
    if base == "char" and stars == 1:
        return "cstr"
    if base == "char" and stars == 2:
        return "ptr(cstr)"
    if base not in scalar:
        return "?" + base
    if stars == 0:
        return scalar[base]
    if stars == 1:
        return "ptr(%s)" % scalar[base]
    return "ptr(ptr(%s))" % scalar[base]


def _flat(c):
    return c[0] if isinstance(c, tuple) else c


def header_prototypes(text):
    """{name: (return class, [argument classes])} of every function declared in the header."""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"^\s*((?:const\s+)?[\w]+(?:\s*\*+)?)\s*(cp2_\w+)\s*\(([^;{]*)\)\s*;", text, flags=re.M):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        args = []
        params = params.strip()
        if params and params != "void":
            for a in params.split(","):
                a = a.strip()
                arr = bool(re.search(r"\[\d*\]\s*$", a))
                a = re.sub(r"\[\d*\]\s*$", "", a)
                # drop the parameter name (last identifier), keep the type with its stars
                mm = re.match(r"^(.*?)(\b\w+)$", a.strip())
                typ = mm.group(1).strip() if mm and mm.group(1).strip() else a
                args.append(_flat(c_type_class(typ, arr)))
        protos[name] = (_flat(c_type_class(ret)), args)
    return protos


def header_config_fields(text):
    m = re.search(r"typedef struct cp2_config \{(.*?)\} cp2_config;", text, flags=re.S)
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    fields = []
    for line in body.split(";"):
        line = line.strip()
        if not line:
            continue
        mm = re.match(r"^(.*?)(\b\w+)$", line)
        fields.append((mm.group(2), _flat(c_type_class(mm.group(1)))))
    return fields


# ---- Nim binding ------------------------------------------------------------------------------------------------
NIM_HANDLES = {"Cp2Ctx": "ctx", "Cp2Dataset": "dataset", "Cp2ProofInput": "proof_input", "Cp2SlotTrees": "slot_trees",
               "Cp2Multi": "multi", "Cp2MultiDataset": "multi_dataset"}


def nim_type_class(t):
    t = re.sub(r"\s+", " ", t.strip())
    scalar = {"cint": "i32", "int32": "i32", "uint32": "u32", "uint64": "u64", "int64": "i64", "csize_t": "usize", "byte": "u8", "uint8": "u8"}
    if t in scalar:
        return scalar[t]
    if t in NIM_HANDLES:
        return "handle:" + NIM_HANDLES[t]
    if t == "cstring":
        return "cstr"
    if t == "pointer":
        return "ptr(void)"
    m = re.match(r"^ptr UncheckedArray\[(\w+)\]$", t)
    if m:
        return "ptr(%s)" % nim_type_class(m.group(1))
    m = re.match(r"^ptr (.+)$", t)
    if m:
        inner = m.group(1)
        if inner == "Cp2Config":
            return "ptr(config)"
        return "ptr(%s)" % nim_type_class(inner)
    return "?" + t


def nim_params(params):
    """'a, b: T; c: U' -> [class of T, class of T, class of U]"""
    out = []
    pending = 0
    for part in re.split(r"[;,]", params):
        part = part.strip()
        if not part:
            continue
        if ":" in part:
            name, typ = part.split(":", 1)
            out += [nim_type_class(typ)] * (pending + 1)
            pending = 0
        else:
            pending += 1
    assert pending == 0, params
    return out


def nim_importc(text):
    """{name: (return class, [argument classes])} of every `{.importc.}` proc."""
    text = strip_nim_comments(text)
    procs = {}
    for m in re.finditer(r"proc\s+(cp2_\w+)\s*\((.*?)\)\s*(?::\s*([^{=]+?))?\s*\{\.\s*importc\s*\.\}", text, flags=re.S):
        name, params, ret = m.group(1), m.group(2), m.group(3)
        procs[name] = (nim_type_class(ret) if ret else "void", nim_params(params))
    return procs


def nim_config_fields(text):
    text = strip_nim_comments(text)
    m = re.search(r"Cp2Config\*\s*\{\.bycopy\.\}\s*=\s*object\n((?:\s+.+\n)+?)\n", text)
    fields = []
    for line in m.group(1).splitlines():
        line = line.strip()
        if not line:
            continue
        names, typ = line.split(":")
        for nme in names.split(","):
            fields.append((nme.strip().rstrip("*"), nim_type_class(typ)))
    return fields


def nim_call_arities(text):
    """[(name, number of arguments)] of every call of a cp2_* function in the Nim text (importc declarations excluded)."""
    text = strip_nim_comments(text)
    text = re.sub(r"proc\s+cp2_\w+\s*\(.*?\{\.\s*importc\s*\.\}", "", text, flags=re.S)
    calls = []
    for m in re.finditer(r"\b(cp2_\w+)\s*\(", text):
        i, depth, commas, empty = m.end(), 1, 0, True
        while depth:
            ch = text[i]
            if ch in "([":
                depth += 1
            elif ch in ")]":
                depth -= 1
            elif ch == "," and depth == 1:
                commas += 1
            if depth and not ch.isspace():
                empty = False
            i += 1
        calls.append((m.group(1), 0 if empty else commas + 1))
    return calls
