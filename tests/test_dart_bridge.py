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
        "tsh_probe_scan_keys", "tsh_probe_batch_keys",  # error-model probes of tests/test_gpu_bands.py
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
    assert len(dart_fields) == len(c_fields) == 19
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
