"""CPU-only tests of the product's host side: the C-ABI library loads and exports every symbol the header
declares, fails loudly without a GPU, and the host logic (getk, signalorder, input checks) mirrors the reference."""
import os
import re

import numpy as np
import pytest


@pytest.fixture(scope="module")
def NMFk():
    import nmfk_jl_amd

    nmfk_jl_amd.build()
    return nmfk_jl_amd


def test_library_exports_every_declared_symbol(NMFk):
    import ctypes

    from importlib import import_module

    _lib = import_module("nmfk_jl_amd._lib")
    text = open(_lib.HEADER_PATH).read()
    declared = sorted(set(re.findall(r"^(?:int|const char \*)\s*(nmfk_[a-z_0-9A-Z]+)\s*\(", text, flags=re.M)))
    assert len(declared) >= 15
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), f"{name} is declared in include/nmfk_hip.h but not exported"
    assert NMFk.lib().nmfk_version() >= 100


def test_no_cpu_fallback(NMFk):
    if NMFk.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(NMFk.NMFkError) as e:
        NMFk.Context(0)
    assert e.value.code == 7 and "no CPU fallback" in str(e.value)
    with pytest.raises(NMFk.NMFkError):
        NMFk.execute(np.ones((4, 3), np.float32), range(2, 3), 2, load=False, save=False)


def test_params_struct_matches_header(NMFk):
    p = NMFk.default_params()
    assert (p.tol, p.tolOF, p.lambda_, p.weight) == (1e-19, 1e-3, 1e-32, 1.0)
    assert (p.maxiter, p.maxreattempts, p.maxbaditers, p.stopconv) == (10000, 2, 10, 1000)
    assert (p.Wfixed, p.Hfixed, p.normalize, p.compute) == (0, 0, 1, 0)
    with pytest.raises(TypeError):
        NMFk.default_params(nonsense=1)


def test_product_never_imports_the_oracle():
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "nmfk.jl_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().lower().replace(
                    "cpu oracle", "").replace("oracle's", "").replace("oracle/nmfk_oracle.c", "").replace(
                    "the oracle", ""), f


def test_getk_matches_reference_rules(NMFk, oracle):
    cases = [(range(2, 6), [0.99, 0.85, -0.57, -0.67], {}), (range(2, 5), [0.1, 0.2, 0.3], {}),
             (range(2, 5), [0.1, 0.2, 0.3], dict(strict=False)), (range(2, 5), [np.nan] * 3, {}), ([3], [0.6], {}),
             ([3], [0.4], {}), ([3], [0.4], dict(strict=False)), (range(2, 5), [0.6, 0.5, 0.7], {}),
             (range(2, 4), [-1, 0.9, 0.2], {}), (range(2, 5), [0.6, np.nan, 0.2], dict(strict=False))]
    for nkrange, rob, kw in cases:
        assert NMFk.getk(nkrange, rob, **kw) == oracle.getk(nkrange, rob, **kw)
    assert NMFk.getk(range(2, 6), [0.99, 0.85, -0.57, -0.67]) == 3


def test_signalorder(NMFk, oracle):
    rng = np.random.default_rng(0)
    W, H = rng.random((9, 4)), rng.random((4, 6))
    so = NMFk.signalorder(W, H)
    assert (so == oracle.signalorder(W, H)).all()
    contrib = [float((W[:, i:i + 1] @ H[i:i + 1, :]).sum()) for i in range(4)]  # Post:153 literally
    assert list(so) == list(np.argsort(-np.array(contrib), kind="stable"))


def test_input_checks(NMFk):
    X = np.ones((5, 4), np.float32)
    # test/test_input_checks.jl family: casefilename defaulting, method aliases are a different solver
    load, save, case, mixture, method, algorithm, cw = NMFk.input_checks(X, True, False, "", "null", "simple", "multdiv", False)
    assert case == "nmfk" and method == "simple"
    for alias in ("multdiv", "multmse", "alspgrad", "nmf", "sparsity", "ipopt"):
        with pytest.raises(NotImplementedError):
            NMFk.input_checks(X, False, False, "", "null", alias, "multdiv", False)
    with pytest.raises(ValueError, match="Unknown method"):
        NMFk.input_checks(X, False, False, "", "null", "bogus", "multdiv", False)
    with pytest.raises(ValueError, match="can be executed for matrices"):
        NMFk.input_checks(np.ones((2, 2, 2)), False, False, "", "null", "simple", "multdiv", False)
    Xn = X.copy()
    Xn[0, 0] = np.nan
    with pytest.warns(UserWarning, match="Simple multiplicative NMF will be performed"):
        out = NMFk.input_checks(Xn, False, False, "", "null", "multdiv", "multdiv", False)
    assert out[4] == "simple"


def test_run_seed_shared_with_oracle(NMFk, oracle):
    for args in [(0, 2, 0), (2021, 5, 9), (2 ** 40, 64, 31)]:
        assert NMFk.run_seed(*args) == oracle.run_seed(*args)


def test_x_hash_sidecar(NMFk, tmp_path):
    """check_x_hash! protocol (Exec:68-93): sidecar written once, mismatch warns, digest depends on values and shape."""
    import warnings

    from nmfk_jl_amd.execute import check_x_hash, hash_sha256_hex

    X = np.arange(12, dtype=np.float32).reshape(3, 4)
    xf = str(tmp_path / "c_x_matrix_3_4.jld")
    h = check_x_hash(X, xf)
    assert open(xf + ".sha256").read().strip() == h == hash_sha256_hex(np.asfortranarray(X))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert check_x_hash(X.copy(), xf) == h
    with pytest.warns(UserWarning, match="hash mismatch"):
        check_x_hash(X + 1, xf)
    assert hash_sha256_hex(X.reshape(4, 3)) != h and hash_sha256_hex(X.astype(np.float64)) != h


def _split_top(s):
    """split a comma-separated list at nesting depth 0"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def test_julia_shim_binds_the_declared_c_abi():
    """julia/NMFkHIP.jl cannot run here (no julia in the image): check statically that every ccall names an entry point
    that include/nmfk_hip.h declares and the library exports, with the same number of arguments and matching scalar
    widths (Cint <-> int, Int64 <-> int64_t, Cdouble <-> double, pointers <-> pointers)."""
    import ctypes
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "nmfk_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for mm in re.finditer(r"\b(?:int|const char \*)\s*(nmfk_\w+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S):
        args = [a for a in _split_top(" ".join(mm.group(2).split())) if a != "void"]
        protos[mm.group(1)] = args
    jl = open(os.path.join(root, "julia", "NMFkHIP.jl")).read()
    lib = ctypes.CDLL(os.path.join(root, "nmfk.jl_amd", "libnmfk_hip.so"))
    calls = list(re.finditer(r"ccall\(\(:(\w+),\s*libnmfk\),\s*(\w+),\s*\(", jl))
    assert len(calls) >= 14 and {"nmfk_multi_create", "nmfk_multi_set_X", "nmfk_multi_sweep", "nmfk_mu_sweep"} <= {c.group(1) for c in calls}
    for mm in calls:
        name = mm.group(1)
        assert name in protos, f"{name} is not declared in include/nmfk_hip.h"
        assert hasattr(lib, name), f"{name} is not exported by libnmfk_hip.so"
        i, depth, start = mm.end(), 1, mm.end()  # argument-type tuple: up to its closing parenthesis
        while depth:
            depth += {"(": 1, ")": -1}.get(jl[i], 0)
            i += 1
        jtypes = _split_top(jl[start:i - 1])
        cargs = protos[name]
        assert len(jtypes) == len(cargs), (name, jtypes, cargs)
        for jt, ca in zip(jtypes, cargs):
            is_ptr = "*" in ca or "[" in ca
            if is_ptr:
                assert jt.startswith(("Ptr{", "Ref{")) or jt == "Cstring", (name, jt, ca)
            elif "uint64_t" in ca:
                assert jt in ("UInt64",), (name, jt, ca)
            elif "int64_t" in ca:
                assert jt in ("Int64", "Clonglong"), (name, jt, ca)
            elif "double" in ca:
                assert jt in ("Cdouble", "Float64"), (name, jt, ca)
            elif re.search(r"\bint\b", ca):
                assert jt in ("Cint", "Int32"), (name, jt, ca)


def _julia_function_kwargs(jl, head):
    """keyword names of the Julia method whose definition starts with `head` (text up to the closing parenthesis)"""
    import re

    i = jl.index(head)
    j, depth = jl.index("(", i), 0
    k = j
    while True:
        depth += {"(": 1, ")": -1}.get(jl[k], 0)
        k += 1
        if depth == 0:
            break
    sig = jl[j + 1:k - 1]
    if ";" not in sig:
        return []
    parts = _split_top(sig[sig.index(";") + 1:])
    return [re.match(r"\s*(\w+)", q).group(1) for q in parts if q.strip() and not q.strip().endswith("...")]


def test_julia_shim_accepts_every_keyword_of_the_reference_signatures():
    """the keyword arguments of execute (range / single k), execute_run, execute_singlerun_compute and
    NMFmultiplicative that reach this path, by NAME (tests/golden/reference_kwargs.json, extracted from the reference
    by tests/golden/make_reference_kwargs.py; re-extracted and compared when /root/reference is present)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fix = os.path.join(root, "tests", "golden", "reference_kwargs.json")
    ref = json.load(open(fix))
    if os.path.isdir("/root/reference/src"):
        out = subprocess.run([sys.executable, os.path.join(root, "tests", "golden", "make_reference_kwargs.py")],
                             capture_output=True, text=True, check=True).stdout
        assert json.loads(out) == ref, "tests/golden/reference_kwargs.json is stale"
    jl = open(os.path.join(root, "julia", "NMFkHIP.jl")).read()
    rng = _julia_function_kwargs(jl, "function execute(X::AbstractArray{T,N}, nkrange::Union{Vector{Int},AbstractUnitRange{Int}}, nNMF::Integer=10;")
    one = _julia_function_kwargs(jl, "function execute(X::AbstractArray{T,N}, nk::Integer, nNMF::Integer=10;")
    run = _julia_function_kwargs(jl, "function execute_run(X::AbstractMatrix{T}, nk::Int, nNMF::Int;")
    many = _julia_function_kwargs(jl, "function execute_many(")
    assert set(ref["execute_range"]) <= set(rng), set(ref["execute_range"]) - set(rng)
    assert set(ref["execute_single"]) <= set(one), set(ref["execute_single"]) - set(one)
    assert set(ref["execute_run"]) <= set(run), set(ref["execute_run"]) - set(run)
    # defaults that decide behaviour: the cache is ON by default, like the reference (Exec:178, 236)
    assert "load::Bool=true, save::Bool=true" in jl.replace("\n", " ").replace("\t", "")
    # execute() forwards execute_run's keywords through execute_many (Exec:304 forwards kw...)
    for name in ("acceptratio", "acceptfactor", "best", "nanaction", "loadall", "saveall", "weight", "veryquiet"):
        assert name in many, name
    # keywords consumed deeper down (Exec:729, Mult:24): peeled out of kw... by name
    consumed = {"tol", "tolOF", "lambda", "maxreattempts", "maxbaditers", "maxiter", "stopconv", "Wfixed", "Hfixed", "Winit", "Hinit",
                "seed", "normalizevector", "weight", "quiet"}
    assert consumed <= set(ref["NMFmultiplicative"]) | {"quiet"}
    for name in consumed - {"weight", "quiet"}:
        assert (":" + name) in jl, f"NMFmultiplicative keyword {name} is not handled by the shim"
    # the rest of execute_singlerun_compute's keywords select other solvers / options that default to off and are
    # rejected loudly or meaningless here
    for name in ("transpose", "deltas", "ratios", "mixture"):
        assert name in run
    assert set(ref["ExecuteOptions"]) == set(_struct_fields(jl, "Base.@kwdef struct ExecuteOptions"))
    # exception types of the reference on this path
    for needle in ('throw(ArgumentError("NMFk analysis can be executed for matrices!"))', "throw(ErrorException(msg))",
                   'error("Input array has a zero dimension!', "@assert size(Winit) == (n, nk)",
                   'error("Initial values for the W matrix entries include NaNs!")'):
        assert needle in jl, needle
    # and the Python mirror accepts the same names
    import inspect

    import nmfk_jl_amd as NMFk

    py = set(inspect.signature(NMFk.execute).parameters)
    assert set(ref["execute_range"]) - {"dims"} <= py and set(ref["execute_single"]) - {"dims"} <= py
    assert set(ref["ExecuteOptions"]) == {f.name for f in __import__("dataclasses").fields(NMFk.ExecuteOptions)}


def _struct_fields(jl, head):
    import re

    body = jl[jl.index(head):]
    body = body[body.index("\n") + 1:body.index("\nend")]
    return [re.match(r"\s*(\w+)", l).group(1) for l in body.split("\n") if l.strip()]


def test_shard_plan_matches_the_python_plan():
    """nmfk_shard_plan (C ABI, host arithmetic) against parallel.plan_shards: every restart owned exactly once, rank g
    owns {g, g + N, ...}, every shard padded to ceil(nruns / N)."""
    import nmfk_jl_amd as NMFk
    from nmfk_jl_amd import _lib

    for nruns in (1, 2, 5, 8, 32, 33):
        for world in (1, 2, 3, 8):
            c, chunks = NMFk.parallel.plan_shards([2, 3], nruns, world)
            seen = []
            for g in range(world):
                cnt, pad = _lib.shard_plan(nruns, world, g)
                assert pad == c
                mine = [rs for q, rs, gg in chunks if gg == g and q == 0]
                assert cnt == (len(mine[0]) if mine else 0)
                seen += list(range(g, nruns, world))[:cnt]
            assert sorted(seen) == list(range(nruns))
    with pytest.raises(NMFk.NMFkError):
        _lib.shard_plan(4, 2, 2)


def test_three_term_bf16_split_is_fp32_accurate():
    """The numerics behind nmfk_step_hyb.hip, modelled in numpy: x = h + m + l with bf16 terms (round to nearest even,
    residuals exact in fp32), inner products from the six term products of weight >= 2^-16 (hh, hm, mh, mm, hl, lh),
    each exact in an fp32 accumulator.  The result is as close to the Float64 inner product as plain fp32 arithmetic."""
    def bf16(x):  # round-to-nearest-even to 8 significand bits, kept as float32
        u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
        return u.astype(np.uint32).view(np.float32)

    def split3(x):
        h = bf16(x)
        r1 = (x - h).astype(np.float32)
        m = bf16(r1)
        r2 = (r1 - m).astype(np.float32)
        return h, m, bf16(r2)

    rng = np.random.default_rng(5)
    k = 16
    a = (rng.random((4096, k)) * rng.choice([1e-3, 1.0, 30.0], size=(4096, 1))).astype(np.float32)
    b = rng.random((4096, k)).astype(np.float32)
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)
    # the residuals of the split are exact: h + m + l reproduces x to the last bf16 rounding of l (<= 2^-25 relative)
    assert np.max(np.abs((ah.astype(np.float64) + am + al) - a) / a) <= 2.0 ** -24
    exact = np.sum(a.astype(np.float64) * b.astype(np.float64), axis=1)
    six = sum(np.sum(x.astype(np.float64) * y.astype(np.float64), axis=1)
              for x, y in ((ah, bh), (ah, bm), (am, bh), (am, bm), (ah, bl), (al, bh)))
    six32 = six.astype(np.float32)  # the accumulator is fp32
    plain32 = np.zeros(4096, dtype=np.float32)
    for c in range(k):
        plain32 = (plain32 + a[:, c] * b[:, c]).astype(np.float32)
    err_six = np.max(np.abs(six32 - exact) / exact)
    err_plain = np.max(np.abs(plain32 - exact) / exact)
    assert err_six <= 2.0 ** -22 and err_six <= 4 * err_plain
    # a two-term split (the usual "bf16x3") would NOT be enough: 2^-16 relative
    two = sum(np.sum(x.astype(np.float64) * y.astype(np.float64), axis=1) for x, y in ((ah, bh), (ah, bm), (am, bh)))
    assert np.max(np.abs(two - exact) / exact) > 2.0 ** -19
