"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol the header declares
(no compute calls - there is no GPU here)."""
import ctypes
import os

import pytest

from tianshou_amd import _lib
from tianshou_amd.build import build_library


@pytest.fixture(scope="module")
def lib():
    build_library()
    return ctypes.CDLL(_lib.LIB_PATH)


def test_header_symbols_exported(lib):
    names = _lib.declared_symbols()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/tsengine.h but not exported: {missing}"


def test_version_and_error_string(lib):
    lib.ts_version.restype = ctypes.c_char_p
    lib.ts_last_error.restype = ctypes.c_char_p
    assert b"gfx950" in lib.ts_version()
    assert isinstance(lib.ts_last_error(), bytes)


def test_argument_validation_without_gpu(lib):
    # validation happens before any HIP call, so these are safe on a GPU-less host
    lib.ts_last_error.restype = ctypes.c_char_p
    rc = lib.ts_gae_scan(None, None, None, None, 0, None, None, None, ctypes.c_int64(0), None,
                         ctypes.c_int64(-1), ctypes.c_double(0.99), ctypes.c_double(0.95),
                         ctypes.c_double(1.0), ctypes.c_double(1.0), None, None, None, None, None, None)
    assert rc == _lib.TS_ERR_INVALID_ARG and b"negative" in lib.ts_last_error()
    rc = lib.ts_nstep_return(None, None, None, None, ctypes.c_int64(4), ctypes.c_int64(1),
                             ctypes.c_int64(0), ctypes.c_int64(8), ctypes.c_double(0.9), None, None, None)
    assert rc == _lib.TS_ERR_INVALID_ARG
    rc = lib.ts_segtree_setitem(None, None, ctypes.c_int64(6), None, None, 1, ctypes.c_int64(1), None)
    assert rc == _lib.TS_ERR_INVALID_ARG and b"power of two" in lib.ts_last_error()


def test_product_path_fails_loudly_without_library(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_product_does_not_import_oracle():
    pkg = os.path.dirname(_lib.__file__)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "ts_oracle" not in text, f


def test_every_entry_point_is_documented_in_integration_md():
    """INTEGRATION.md's table names every symbol include/tsengine.h declares (so the reference-side binding list is complete)."""
    import os

    from tianshou_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    expand = text.replace("`ts_profile_begin/end`", "`ts_profile_begin` `ts_profile_end`")
    missing = [s for s in _lib.declared_symbols() if s not in expand]
    assert not missing, missing


def test_python_call_sites_pass_as_many_arguments_as_the_header_declares():
    """ctypes calls without argtypes do not check arity: a wrapper that passes one argument too few marshals garbage into
    the last parameter.  Static check: every `<lib>.ts_*(...)` call in tianshou_amd/ has exactly the number of positional
    arguments of its prototype in include/tsengine.h (a `*self._dims()` argument counts as the length of the tuple that
    method returns)."""
    import ast
    import os
    import re

    from tianshou_amd import _lib

    text = re.sub(r"/\*.*?\*/", "", open(_lib.HEADER_PATH).read(), flags=re.S)
    protos = {}
    for name, params in re.findall(r"\b(ts_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        params = params.strip()
        protos[name] = 0 if params in ("", "void") else params.count(",") + 1
    assert len(protos) >= 80 and protos["ts_version"] == 0 and protos["ts_polyak_update"] == 5

    pkg = os.path.dirname(_lib.__file__)
    checked = 0
    for fn in sorted(os.listdir(pkg)):
        if not fn.endswith(".py"):
            continue
        tree = ast.parse(open(os.path.join(pkg, fn)).read())
        dims_len = {}                                   # class name -> len of the tuple `_dims` returns
        for cls in [n for n in ast.walk(tree) if isinstance(n, ast.ClassDef)]:
            for m in cls.body:
                if isinstance(m, ast.FunctionDef) and m.name == "_dims":
                    ret = [r for r in ast.walk(m) if isinstance(r, ast.Return)][0].value
                    assert isinstance(ret, ast.Tuple)
                    dims_len[cls.name] = len(ret.elts)
            for call in [n for n in ast.walk(cls) if isinstance(n, ast.Call)]:
                call._cls = cls.name                    # noqa: SLF001 - remember the enclosing class for star-args
        for call in [n for n in ast.walk(tree) if isinstance(n, ast.Call)]:
            f = call.func
            if not (isinstance(f, ast.Attribute) and f.attr in protos) or call.keywords:
                continue
            n = 0
            for a in call.args:
                if isinstance(a, ast.Starred):
                    v = a.value
                    ok = isinstance(v, ast.Call) and isinstance(v.func, ast.Attribute) and v.func.attr == "_dims"
                    assert ok and getattr(call, "_cls", None) in dims_len, f"{fn}:{call.lineno}: unsupported star argument"
                    n += dims_len[call._cls]
                else:
                    n += 1
            assert n == protos[f.attr], f"{fn}:{call.lineno}: {f.attr} takes {protos[f.attr]} arguments, call passes {n}"
            checked += 1
    assert checked >= 60, checked


def test_ctypes_structures_match_the_header_structs():
    """Every `typedef struct ts_* { ... }` of include/tsengine.h against its ctypes mirror in the package: the same number of
    members with the same sizes in the same order (a renamed / added / reordered member on one side only would shift
    every later one -- nothing else would notice until a kernel read a wrong hyper-parameter)."""
    import ctypes as C
    import re

    from tianshou_amd import _lib, distq, dqn, drqn, npg, redq, sac, td3

    mirrors = {"ts_ppo_hparams": _lib.PPOHParams, "ts_dqn_hparams": dqn.DQNHParams, "ts_distq_hparams": distq.DistQHParams,
               "ts_rows_replay": drqn.RowsReplay, "ts_npg_hparams": npg.NPGHParams, "ts_sac_hparams": sac.SACHParams,
               "ts_sac_state": sac.SACStateC, "ts_sac_replay": sac.SACReplayC, "ts_redq_state": redq.REDQStateC, "ts_td3_hparams": td3.TD3HParams,
               "ts_td3_state": td3.TD3StateC, "ts_net_desc": _lib.NetDesc, "ts_frame_replay": _lib.FrameReplay}
    text = re.sub(r"/\*.*?\*/", "", open(_lib.HEADER_PATH).read(), flags=re.S)
    sizes = {"double": 8, "float": 4, "int64_t": 8, "uint64_t": 8, "int32_t": 4, "int": 4, "uint8_t": 1}
    seen = set()
    for body, name in re.findall(r"typedef\s+struct\s+ts_\w+\s*\{(.*?)\}\s*(ts_\w+)\s*;", text, flags=re.S):
        if name == "ts_scatter_key":                      # built inline in buffer.py
            continue
        assert name in mirrors, f"{name}: no ctypes mirror registered in this test"
        seen.add(name)
        want = []
        for decl in [d.strip() for d in body.split(";") if d.strip()]:
            base = re.match(r"(?:const\s+)?(\w+)", decl).group(1)
            for item in decl[decl.index(base) + len(base):].split(","):
                item = item.strip()
                count = 1
                arr = re.match(r"(.*?)\[(\w+)\]$", item)          # fixed-size array member: name[N] / name[SOME_DEFINE]
                if arr:
                    item, n = arr.group(1).strip(), arr.group(2)
                    count = int(n) if n.isdigit() else int(re.search(rf"#define\s+{n}\s+(\d+)", text).group(1))
                want.append((item.lstrip("* ").strip(), count * (8 if "*" in item else sizes[base])))        # pointers: 8 bytes
        got = [(f[0], C.sizeof(f[1])) for f in mirrors[name]._fields_]
        assert [w[1] for w in want] == [g[1] for g in got], (name, want, got)
        assert len(want) == len(got)
        # members carry the same names on both sides, except where the Python side groups pointers under its own names
        if not name.endswith("_state"):
            assert [w[0] for w in want] == [g[0] for g in got], (name, want, got)
    assert seen == set(mirrors)
