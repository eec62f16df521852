"""CPU: the `dart:ffi` bridge (tostore_amd/dart/tostore_hip_bridge.dart) against include/tostore_hip.h.
No Dart SDK exists in the image, so the bridge cannot be compiled here; what a compiler would not catch
anyway -- a native typedef whose arity or argument widths differ from the C prototype (dart:ffi binds
by NAME only) -- is checked by parsing both files: every `typedef _XxxC = Ret Function(args)`, the
symbol it is looked up as, its Dart-side twin `_XxxD`, and the TshNghInfo struct.  The ctypes table of
tostore_amd/_ffi.py is held to the same header by the same parser."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tostore_hip.h")
BRIDGE = os.path.join(ROOT, "tostore_amd", "dart", "tostore_hip_bridge.dart")

# C parameter / return type -> dart:ffi native type
C2DART = {
    "void": "Void", "int32_t": "Int32", "int64_t": "Int64", "double": "Double", "float": "Float",
    "char*": "Pointer<Utf8>", "float*": "Pointer<Float>", "double*": "Pointer<Double>",
    "int32_t*": "Pointer<Int32>", "int64_t*": "Pointer<Int64>", "uint8_t*": "Pointer<Uint8>",
    "void*": "Pointer<Void>", "tsh_index*": "Pointer<Void>", "tsh_index**": "Pointer<Pointer<Void>>",
    "tsh_comm*": "Pointer<Void>", "tsh_comm**": "Pointer<Pointer<Void>>",
    "tsh_shard_stream*": "Pointer<Void>", "tsh_shard_stream**": "Pointer<Pointer<Void>>",
    "tsh_mask*": "Pointer<Void>", "tsh_mask**": "Pointer<Pointer<Void>>",
    "tsh_ngh_info*": "Pointer<TshNghInfo>", "tsh_counters*": "Pointer<TshCounters>",
    "tsh_comm_timeline*": "Pointer<TshCommTimeline>",
    "tsh_allgather_fn": "Pointer<NativeFunction<TshAllgatherNative>>",
}
NATIVE2DART = {"Int32": "int", "Int64": "int", "Double": "double", "Float": "double", "Void": "void"}
C2CTYPES = {
    "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "double": ctypes.c_double,
    "char*": ctypes.c_char_p, "float*": ctypes.POINTER(ctypes.c_float), "double*": ctypes.POINTER(ctypes.c_double),
    "int32_t*": ctypes.POINTER(ctypes.c_int32), "int64_t*": ctypes.POINTER(ctypes.c_int64),
    "uint8_t*": ctypes.POINTER(ctypes.c_uint8), "void*": ctypes.c_void_p, "tsh_index*": ctypes.c_void_p,
    "tsh_index**": ctypes.POINTER(ctypes.c_void_p), "tsh_comm*": ctypes.c_void_p,
    "tsh_comm**": ctypes.POINTER(ctypes.c_void_p), "tsh_allgather_fn": ctypes.c_void_p,
    "tsh_shard_stream*": ctypes.c_void_p, "tsh_shard_stream**": ctypes.POINTER(ctypes.c_void_p),
    "tsh_mask*": ctypes.c_void_p, "tsh_mask**": ctypes.POINTER(ctypes.c_void_p),
}


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _ctype(decl):
    """'const float *rows' -> 'float*';  'tsh_index **out' -> 'tsh_index**';  'int32_t k' -> 'int32_t'."""
    decl = decl.replace("const", " ").replace("struct", " ").strip()
    stars = decl.count("*")
    words = decl.replace("*", " ").split()
    base = words[0] if len(words) == 1 or stars or words[0] in ("void",) else words[0]
    return base + "*" * stars


def c_prototypes():
    text = _strip_comments(open(HEADER).read())
    protos = {}
    for ret, name, args in re.findall(r"\b(int32_t|int64_t|void)\s+(tsh_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", text):
        args = args.strip()
        params = [] if args in ("", "void") else [_ctype(a) for a in args.split(",")]
        protos[name] = (ret, params)
    return protos


def c_struct(name):
    text = _strip_comments(open(HEADER).read())
    body = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), text, flags=re.S).group(1)
    return [(t, f) for t, f in re.findall(r"\b(int32_t|int64_t|double|float|uint32_t|uint64_t)\s+(\w+)\s*;", body)]


def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return [re.sub(r"\s+", "", a) for a in out]


def dart_typedefs():
    text = _strip_comments(open(BRIDGE).read())
    tds = {}
    for name, ret, args in re.findall(r"typedef\s+(_\w+)\s*=\s*([\w<>]+)\s+Function\(([^;]*?)\)\s*;", text, flags=re.S):
        tds[name] = (ret, _split_args(args))
    return tds


def dart_lookups():
    text = _strip_comments(open(BRIDGE).read())
    return re.findall(r"lookupFunction<\s*(_\w+)\s*,\s*(_\w+)\s*>\(\s*'(tsh_\w+)'\s*\)", text)


def test_header_parser_sees_every_prototype():
    text = _strip_comments(open(HEADER).read())
    names = sorted(set(re.findall(r"\b(tsh_[a-z_0-9]+)\s*\(", text)))
    protos = c_prototypes()
    assert sorted(protos) == names and len(names) >= 28
    assert protos["tsh_search"] == ("int32_t", ["tsh_index*", "float*", "int32_t", "int32_t", "double", "uint8_t*",
                                                "int64_t*", "double*", "int32_t*"])
    assert protos["tsh_index_create"][1][-1] == "tsh_index**"


def test_dart_native_typedefs_match_the_c_prototypes():
    protos, tds, looks = c_prototypes(), dart_typedefs(), dart_lookups()
    assert len(looks) >= 33
    bound = set()
    for c_td, d_td, sym in looks:
        assert sym in protos, f"{sym} is looked up by the bridge but not declared in tostore_hip.h"
        assert c_td in tds and d_td in tds, f"{sym}: typedef {c_td} / {d_td} missing"
        ret, params = protos[sym]
        n_ret, n_args = tds[c_td]
        want = [C2DART[p] for p in params]
        # an info pointer may be bound untyped
        n_args_norm = [("Pointer<TshNghInfo>" if (w == "Pointer<TshNghInfo>" and a == "Pointer<Void>") else a)
                       for a, w in zip(n_args, want)] + n_args[len(want):]
        assert n_ret == C2DART[ret], f"{sym}: native return {n_ret}, header {ret}"
        assert n_args_norm == want, f"{sym}: native args {n_args} != header {params}"
        # the Dart-side twin: same shape, integers -> int, floating -> double, pointers unchanged
        d_ret, d_args = tds[d_td]
        assert d_ret == NATIVE2DART.get(n_ret, n_ret), f"{sym}: Dart return {d_ret}"
        assert d_args == [NATIVE2DART.get(a, a) for a in n_args], f"{sym}: Dart args {d_args} vs native {n_args}"
        bound.add(sym)
    # every symbol of the header is either bound or on this list of entries a Dart host has no use for
    not_for_dart = {
        "tsh_bench_scan", "tsh_bench_batch",  # measurement hooks of bench.py
        "tsh_probe_scan_keys", "tsh_probe_batch_keys", "tsh_probe_batch_row_band",  # error-model probes of tests/test_gpu_bands.py
        "tsh_index_append_device",            # takes a device pointer: bulk loaders that already hold the column in HBM
    }
    missing = set(protos) - bound - not_for_dart
    assert not missing, f"the bridge binds neither of {sorted(missing)} (bind them or list them as not for Dart)"
    assert not (bound & not_for_dart)
    # the asynchronous and sharded entries the reference's yield contract / BASELINE.json's C4 need
    for must in ("tsh_search_submit", "tsh_search_ready", "tsh_search_wait", "tsh_get_counters", "tsh_search_sharded",
                 "tsh_comm_create", "tsh_comm_unique_id", "tsh_index_create_shard"):
        assert must in bound


def test_dart_counters_struct_matches_the_header():
    text = _strip_comments(open(BRIDGE).read())
    body = re.search(r"final\s+class\s+TshCounters\s+extends\s+Struct\s*\{(.*?)\n\}", text, flags=re.S).group(1)
    dart_fields = re.findall(r"@(\w+)\(\)\s*external\s+(\w+)\s+(\w+)\s*;", body)
    c_fields = c_struct("tsh_counters")
    assert len(dart_fields) == len(c_fields) >= 15
    snake = lambda s: re.sub(r"([A-Z])", lambda m: "_" + m.group(1).lower(), s)  # noqa: E731
    for (ann, dtype, dname), (ctype, cname) in zip(dart_fields, c_fields):
        assert ann == C2DART[ctype], f"{cname}: @{ann} vs {ctype}"
        assert dtype == NATIVE2DART[ann]
        assert snake(dname) == cname, f"field order: Dart {dname} vs C {cname}"


def test_dart_timeline_struct_matches_the_header():
    text = _strip_comments(open(BRIDGE).read())
    body = re.search(r"final\s+class\s+TshCommTimeline\s+extends\s+Struct\s*\{(.*?)\n\}", text, flags=re.S).group(1)
    dart_fields = re.findall(r"@(\w+)\(\)\s*external\s+(\w+)\s+(\w+)\s*;", body)
    c_fields = c_struct("tsh_comm_timeline")
    assert len(dart_fields) == len(c_fields) == 20
    snake = lambda s: re.sub(r"([A-Z])", lambda m: "_" + m.group(1).lower(), s)  # noqa: E731
    for (ann, dtype, dname), (ctype, cname) in zip(dart_fields, c_fields):
        assert ann == C2DART[ctype], f"{cname}: @{ann} vs {ctype}"
        assert dtype == NATIVE2DART[ann]
        assert snake(dname) == cname.replace("d2h", "d2h"), f"field order: Dart {dname} vs C {cname}"


def test_bridge_checks_the_abi_version_of_the_header():
    hdr = int(re.search(r"#define\s+TSH_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    dart = int(re.search(r"static\s+const\s+int\s+abiVersion\s*=\s*(\d+)", open(BRIDGE).read()).group(1))
    from tostore_amd import _ffi

    assert hdr == dart == _ffi.ABI_VERSION


def test_dart_info_struct_matches_the_header():
    text = _strip_comments(open(BRIDGE).read())
    body = re.search(r"final\s+class\s+TshNghInfo\s+extends\s+Struct\s*\{(.*?)\n\}", text, flags=re.S).group(1)
    dart_fields = re.findall(r"@(\w+)\(\)\s*external\s+(\w+)\s+(\w+)\s*;", body)
    c_fields = c_struct("tsh_ngh_info")
    assert len(dart_fields) == len(c_fields) >= 15
    snake = lambda s: re.sub(r"([A-Z])", lambda m: "_" + m.group(1).lower(), s)  # noqa: E731
    for (ann, dtype, dname), (ctype, cname) in zip(dart_fields, c_fields):
        assert ann == C2DART[ctype], f"{cname}: @{ann} vs {ctype}"
        assert dtype == NATIVE2DART[ann]
        assert snake(dname) == cname, f"field order: Dart {dname} vs C {cname}"


def test_ctypes_table_matches_the_c_prototypes():
    from tostore_amd import _ffi

    protos = c_prototypes()
    assert sorted(protos) == sorted(_ffi.SIGNATURES)
    for name, (ret, params) in protos.items():
        res, args = _ffi.SIGNATURES[name]
        assert res is C2CTYPES[ret], f"{name}: restype"
        assert len(args) == len(params), f"{name}: {len(args)} ctypes args, header has {len(params)}"
        for i, (a, p) in enumerate(zip(args, params)):
            if p in ("tsh_ngh_info*", "tsh_counters*", "tsh_comm_timeline*"):
                assert issubclass(a, ctypes._Pointer) and issubclass(a._type_, ctypes.Structure), f"{name} arg {i}"
            else:
                want = C2CTYPES[p]
                same = a is want or (a is ctypes.c_void_p and want is ctypes.c_void_p)
                # POINTER(c_int32) objects are cached by ctypes, so identity holds for them too
                assert same, f"{name} arg {i}: ctypes {a} vs header {p}"
    # struct mirrors: field order and widths
    for cname, cls in (("tsh_counters", _ffi.TshCounters), ("tsh_ngh_info", _ffi.TshNghInfo),
                       ("tsh_comm_timeline", _ffi.TshCommTimeline)):
        cf = c_struct(cname)
        assert [f for _, f in cf] == [f for f, _ in cls._fields_], cname
        for (ct, _), (_, pt) in zip(cf, cls._fields_):
            assert ctypes.sizeof(pt) == {"int32_t": 4, "int64_t": 8, "double": 8, "float": 4}[ct]


# ---- the VectorIndexManager hook (tostore_amd/dart/hip_vector_hook.dart) --------------------
HOOK = os.path.join(ROOT, "tostore_amd", "dart", "hip_vector_hook.dart")
REF_VIM = "/root/reference/lib/src/core/vector_index_manager.dart"


def _call_args(text, start):
    """Arguments of the call whose '(' is at text[start]: (positional count, set of named arguments)."""
    depth, i, cur, args = 0, start, "", []
    while True:
        ch = text[i]
        if ch in "([{<" and not (ch == "<" and text[i - 1] == " "):
            depth += 1
        elif ch in ")]}>" and not (ch == ">" and text[i - 1] in " ="):
            depth -= 1
            if depth == 0:
                break
        if ch == "," and depth == 1:
            args.append(cur.strip())
            cur = ""
        elif depth >= 1 and not (depth == 1 and ch == "("):
            cur += ch
        i += 1
    if cur.strip():
        args.append(cur.strip())
    named = {a.split(":")[0].strip() for a in args if re.match(r"^\w+\s*:", a)}
    return len(args) - len(named), named


def _bridge_members():
    """name -> (positional parameter count, named parameter names) of HipVectorBackend's methods, plus its getters."""
    text = _strip_comments(open(BRIDGE).read())
    cls = text[text.index("final class HipVectorBackend"):text.index("final class HipShardComm")]
    members = {}
    for m in re.finditer(r"\n  (?:static\s+)?[\w<>?, ]+?\s+(\w+)\(([^)]*)\)\s*(?:async\s*)?(?:\{|=>)", cls):
        name, params = m.group(1), m.group(2)
        pos, _, named = params.partition("{")
        npos = len([p for p in pos.split(",") if p.strip()])
        nnames = {p.strip().split("=")[0].split()[-1] for p in named.rstrip("} ").split(",") if p.strip()}
        members[name] = (npos, nnames)
    getters = set(re.findall(r"\n  (?:static\s+)?[\w<>?]+\s+get\s+(\w+)", cls))
    return members, getters


def test_hook_calls_only_what_the_bridge_has():
    """Every HipVectorBackend member the hook uses exists in the bridge with the arity / named parameters used."""
    hook = _strip_comments(open(HOOK).read())
    members, getters = _bridge_members()
    assert {"tryOpen", "search", "searchAsync", "append", "setDeleted", "dispose", "counters"} <= set(members)
    used = 0
    for m in re.finditer(r"\b(?:hip|b|HipVectorBackend|e\.value)\.(\w+)(\()?", hook):
        name, is_call = m.group(1), m.group(2) is not None
        if not is_call:
            assert name in getters or name in members, "the hook reads HipVectorBackend.%s, which the bridge lacks" % name
            continue
        assert name in members, "the hook calls HipVectorBackend.%s(), which the bridge lacks" % name
        npos, named = _call_args(hook, m.end() - 1)
        want_pos, want_named = members[name]
        assert npos == want_pos, "%s: the hook passes %d positional arguments, the bridge takes %d" % (name, npos, want_pos)
        assert named <= want_named, "%s: named arguments %s not in the bridge's %s" % (name, sorted(named), sorted(want_named))
        used += 1
    assert used >= 8
    # the mixin's own surface: the four call sites the maintainer wires up
    for fn in ("hipSearch", "hipAfterInsert", "hipAfterDelete", "hipDrop", "hipDropTable", "hipDisposeAll", "hipFor"):
        assert re.search(r"\b%s\(" % fn, hook), fn
    assert "mixin HipVectorHook implements HipVectorHookHost" in hook


def test_hook_call_sites_name_real_reference_lines():
    """The line numbers the hook cites hold what it says (checked where the reference tree exists: this container)."""
    import pytest

    if not os.path.exists(REF_VIM):
        pytest.skip("no reference tree on this machine")
    ref = open(REF_VIM).read().split("\n")

    def span(a, b):
        return "\n".join(ref[a - 1:b])

    assert "class VectorIndexManager" in span(28, 28)
    assert "_metaLoadingFutures" in span(35, 37)
    assert "_prepareInsertVectorsBatch" in span(349, 356) and "startNodeId = meta.nextNodeId" in span(359, 359)
    assert "_graphEngine.insertBatch" in span(368, 377) and "dirtyRawVectorPages" in span(378, 388)
    assert "_persistMeta" in span(401, 401) and "_graphEngine.deleteBatch" in span(429, 434)
    assert "_toFloat32" in span(514, 520) and "_normalizeFloat32" in span(514, 520)
    assert "_graphEngine.search(" in span(536, 551) and "lease?.release()" in span(536, 551)
    assert "return true;" in span(1154, 1158)
    # the ef cap the hook can bring back (hipHonourEfCap): the lines it restates
    eng = open(os.path.join(os.path.dirname(REF_VIM), "ngh_graph_engine.dart")).read().split("\n")
    assert "meta.medoidNodeId < 0) return const []" in eng[78 - 1]
    assert "efSearch ?? meta.efSearch" in eng[80 - 1] and "min(efRaw, max(topK * 5, 32))" in eng[82 - 1]
    meta = open(os.path.join(os.path.dirname(os.path.dirname(REF_VIM)), "model", "ngh_index_meta.dart")).read().split("\n")
    assert "efSearch = 64" in meta[196 - 1]
    # ... and the hook file cites exactly these
    hook = open(HOOK).read()
    assert "hipHonourEfCap" in hook and "efSearch: efSearch" in hook and "meta.medoidNodeId < 0" in hook
    assert "_hipAsyncAboveFloats = 1 << 28" in hook  # (256 M floats: what its comment derives)
    for cite in (":349-356", ":359,", ":368-388", ":378-388", ":401)", ":429-434", ":514-520", ":536-551", ":1154-1158", ":1192", ":1198", ":1205"):
        assert cite in hook, cite
    assert "void clearCacheForTable" in span(1192, 1192) and "void clearCacheForIndex" in span(1198, 1198)
    assert "Future<void> dispose()" in span(1205, 1205)
