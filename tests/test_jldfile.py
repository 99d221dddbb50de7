"""nmfk.jl_amd/jldfile.py against JLD files written by Julia itself (tests/golden/julia_written_*.jld) and in round trips."""
import os

import numpy as np
import pytest

from nmfk_jl_amd import jldfile, resultio  # noqa: E402  (conftest puts the repository root on sys.path)

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_reads_julia_written_files():
    for name, n in (("julia_written_Hmatrix-2-2_10-1000.jld", 10), ("julia_written_Wmatrix-4-4_100-1000.jld", 100)):
        path = os.path.join(GOLD, name)
        assert open(path, "rb").read(len(jldfile.MAGIC)) == jldfile.MAGIC
        root, R = jldfile.read_structure(path)
        top = root["children"]
        assert {"assignments", "best_silhouettes", "_creator", "_types", "_refs"} <= set(top)
        cr = {k: int(v["value"]) for k, v in top["_creator"]["children"].items()}
        assert cr["WORD_SIZE"] == 64 and cr["ENDIAN_BOM"] == 0x04030201 and cr["JULIA_MAJOR"] == 1
        sil = top["best_silhouettes"]["value"]
        assert sil.dtype == np.float64 and sil.shape == (n,) and np.all(np.abs(sil) <= 1)
        tag = top["_types"]["children"]["00000001"]["attrs"]["julia type"]
        assert tag.startswith("Clustering.KmeansResult{Core.Array{Core.Float64,2}")
        km = top["assignments"]["value"]  # the KmeansResult struct: references into /_refs
        byhdr = {c["header"]: c for c in top["_refs"]["children"].values()}
        assign = byhdr[km["assignments_"][0]]["value"]
        k = int(name.split("-")[1])
        assert assign.shape == (n,) and set(assign) <= set(range(1, k + 1))
        centers = byhdr[km["centers_"][0]]["value"]
        assert centers.ndim == 2 and centers.shape[1] == k  # d x k, Julia's column-major order kept
        assert km["iterations_"] >= 1 and km["converged_"] in (0, 1)
        flat = jldfile.load(path)
        assert "best_silhouettes" in flat and "_creator" not in flat


def _messages(R, node):
    return [(t, bytes(b)) for t, b in R.messages(node["header"])]


def test_writer_uses_the_structures_julia_writes(tmp_path):
    """A Float64 vector written here carries the same message types as Julia's `best_silhouettes`, with byte-identical
    dataspace and datatype messages; groups are link-message groups like Julia's; the user block carries the JLD magic."""
    ref_root, Rr = jldfile.read_structure(os.path.join(GOLD, "julia_written_Hmatrix-2-2_10-1000.jld"))
    sil = ref_root["children"]["best_silhouettes"]
    fn = str(tmp_path / "x.jld")
    jldfile.save(fn, {"best_silhouettes": sil["value"], "n": np.int64(64)})
    root, R = jldfile.read_structure(fn)
    mine, theirs = _messages(R, root["children"]["best_silhouettes"]), _messages(Rr, sil)
    strip = lambda ms: [(t, b) for t, b in ms if t != 0]  # Julia's headers are padded with NIL messages
    assert [t for t, _ in strip(mine)] == [t for t, _ in strip(theirs)] == [0x0001, 0x0003, 0x0005, 0x0008, 0x0012]
    for i in (0, 1, 2):  # dataspace, datatype, fill value: byte for byte
        assert strip(mine)[i][1] == strip(theirs)[i][1]
    assert strip(mine)[3][1][:4] == strip(theirs)[3][1][:4]  # layout: version 3, compact, 80 bytes
    np.testing.assert_array_equal(root["children"]["best_silhouettes"]["value"], sil["value"])
    word = ref_root["children"]["_creator"]["children"]["WORD_SIZE"]
    assert strip(_messages(R, root["children"]["n"]))[1][1] == strip(_messages(Rr, word))[1][1]  # Int64 datatype message
    kinds = lambda RR, node: sorted({t for t, _ in RR.messages(node["header"])} - {0, 0x10})
    assert kinds(R, root) == kinds(Rr, ref_root) == [0x0002, 0x0006, 0x000A]
    assert open(fn, "rb").read(512).startswith(jldfile.MAGIC) and open(fn, "rb").read()[512:520] == jldfile.SIG


@pytest.mark.parametrize("n,m,k", [(15, 5, 3), (3000, 512, 16)])
def test_result_file_roundtrip(tmp_path, n, m, k):
    rng = np.random.default_rng(1)
    W, H = rng.random((n, k), dtype=np.float32), rng.random((k, m), dtype=np.float32)
    fn = str(tmp_path / f"case_{n}_{m}_{k}_10.jld")
    resultio.save(fn, W=W, H=H, fit=np.float32(1.25), robustness=np.float32(0.5), aic=np.float32(-3.0))
    z = resultio.load(fn)
    assert set(z) == {"W", "H", "fit", "robustness", "aic"}  # the keys of src/NMFkExecute.jl:325
    np.testing.assert_array_equal(z["W"], W)
    np.testing.assert_array_equal(z["H"], H)
    assert z["W"].dtype == np.float32 and float(z["fit"]) == 1.25 and float(z["aic"]) == -3.0
    root, _ = jldfile.read_structure(fn)
    assert root["children"]["W"]["shape"] == (k, n)  # HDF5 dimensions = Julia's reversed
    assert not [f for f in os.listdir(tmp_path) if ".tmp" in f]


def test_all_payload_roundtrip(tmp_path):
    """the 12 variables of the -all.jld payload (src/NMFkExecute.jl:652), Vector{Matrix} as reference datasets"""
    rng = np.random.default_rng(2)
    R, n, m, k = 4, 20, 6, 3
    Ws, Hs = [rng.random((n, k), dtype=np.float32) for _ in range(R)], [rng.random((k, m), dtype=np.float32) for _ in range(R)]
    payload = {"W": Ws, "H": Hs, "Wmean": Ws[0], "Hmean": Hs[0], "Wvar": Ws[1], "Hvar": Hs[1], "Wbest": Ws[2], "Hbest": Hs[2],
               "fit": np.arange(R, dtype=np.float32), "Cluster Silhouettes": np.ones(k, np.float32),
               "Cluster assignments": np.tile(np.arange(1, k + 1, dtype=np.int64)[:, None], (1, R)),
               "Cluster centroids": Hs[3]}
    fn = str(tmp_path / "c-all.jld")
    resultio.save(fn, **payload)
    z = resultio.load(fn)
    assert list(z) == list(payload)
    for a, b in zip(z["W"], Ws):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(z["Cluster assignments"], payload["Cluster assignments"])
    root, _ = jldfile.read_structure(fn)
    assert root["children"]["W"]["attrs"]["julia type"] == "Core.Array{Core.Array{Core.Float32,2},1}"
    assert len(root["children"]["_refs"]["children"]) == 2 * R
    w, h, f = resultio.load(fn, "W", "H", "fit").values()  # JLD.load(filename, "W", "H", "fit") of Exec:503
    assert len(w) == R and f.shape == (R,)


def test_superblock_fields_match_the_julia_written_files(tmp_path):
    """the superblock's end-of-file address is ABSOLUTE.  libhdf5 limits addresses to eof - base, so a
    file that stores (size - 512) puts its last 512 bytes -- where the root group header sits -- out of reach and
    JLD.load fails with an address overflow although our own reader (which ignored the field) was happy.  Pin base and EOF
    of a written file to what both Julia-written fixtures hold: base = 512, EOF = file size; the reader rejects a
    file whose EOF differs from its size, except the relative value older versions of this package wrote (warning)."""
    import struct

    from nmfk_jl_amd import jldfile

    def fields(path):
        b = open(path, "rb").read()
        at = b.find(jldfile.SIG)
        base, _, eof, _ = struct.unpack_from("<QQQQ", b, at + 24)
        return at, base, eof, len(b)

    for name in ("julia_written_Hmatrix-2-2_10-1000.jld", "julia_written_Wmatrix-4-4_100-1000.jld"):
        at, base, eof, size = fields(os.path.join(GOLD, name))
        assert (at, base, eof) == (512, 512, size)
    fn = str(tmp_path / "w.jld")
    jldfile.save(fn, {"W": [np.ones((3, 2), np.float32)] * 2, "fit": np.arange(4, dtype=np.float32), "big": np.zeros((64, 64))})
    at, base, eof, size = fields(fn)
    assert (at, base, eof) == (512, 512, size) == (512, 512, os.path.getsize(fn))
    assert set(jldfile.load(fn)) >= {"W", "fit", "big"}
    raw = bytearray(open(fn, "rb").read())
    struct.pack_into("<Q", raw, 512 + 40, size - 512)  # the round-2 defect: files this package wrote before the fix
    open(fn, "wb").write(bytes(raw))
    with pytest.warns(UserWarning, match="relative to the 512-byte user block"):  # (still readable, with a warning)
        assert set(jldfile.load(fn)) >= {"W", "fit", "big"}
    struct.pack_into("<Q", raw, 512 + 40, size - 100)  # any other value: truncated or mis-written
    open(fn, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="end-of-file"):
        jldfile.load(fn)


def test_kmeans_result_struct_is_written_like_julia(tmp_path):
    """The robustkmeans cache of the reference (src/NMFkCluster.jl:236-244: JLD.save(filename, "assignments", sc,
    "best_silhouettes", ...)) holds a Clustering.KmeansResult struct.  The writer's committed compound datatype must be,
    byte for byte, the datatype message of the Julia-written fixture; a written file reads back like the fixture does."""
    import struct

    from nmfk_jl_amd import jldfile

    gold = os.path.join(GOLD, "julia_written_Hmatrix-2-2_10-1000.jld")
    root, R = jldfile.read_structure(gold)
    tnode = root["children"]["_types"]["children"]["00000001"]
    assert tnode["attrs"]["julia type"] == jldfile.KMEANS_JULIA_TYPE
    gold_msg = next(bytes(b) for t, b in R.messages(tnode["header"]) if t == 0x0003)
    mine = jldfile.kmeans_result_datatype()
    assert gold_msg[:len(mine)] == mine and not any(gold_msg[len(mine):])  # (the stored message is padded to 8 bytes)
    g = jldfile.load(gold)
    assert g["assignments"]["assignments_"].tolist() == [1, 2, 1, 1, 1, 1, 2, 1, 1, 1] and g["assignments"]["counts_"].tolist() == [8, 2]
    assert g["assignments"]["centers_"].shape == (2, 2) and g["assignments"]["converged_"] == 1
    # write the same content, read it back, and compare the object-level structure with the fixture's
    fn = str(tmp_path / "Hmatrix-2-2_10-1000.jld")
    jldfile.save(fn, {"assignments": g["assignments"], "best_silhouettes": g["best_silhouettes"]})
    z = jldfile.load(fn)
    for key, val in g["assignments"].items():
        assert np.array_equal(np.asarray(z["assignments"][key]), np.asarray(val)), key
    assert np.array_equal(z["best_silhouettes"], g["best_silhouettes"])
    r2, R2 = jldfile.read_structure(fn)
    a, b = root["children"], r2["children"]
    assert a["assignments"]["type"] == b["assignments"]["type"] and a["assignments"]["shape"] == b["assignments"]["shape"]
    assert b["_types"]["children"]["00000001"]["attrs"] == tnode["attrs"]
    assert {k: (v["type"], v["shape"]) for k, v in a["_refs"]["children"].items()} == \
           {k: (v["type"], v["shape"]) for k, v in b["_refs"]["children"].items()}
    # the dataset's datatype message is the SHARED form pointing at the committed type, as in the fixture
    for Rx, node, types in ((R, a["assignments"], a["_types"]), (R2, b["assignments"], b["_types"])):
        o = Rx.at(node["header"])
        p = o + 16
        for _ in range(struct.unpack_from("<H", Rx.buf, o + 2)[0]):
            mtype, msize, flags = struct.unpack_from("<HHB", Rx.buf, p)
            if mtype == 3:
                assert flags == 3 and Rx.buf[p + 8:p + 10] == b"\x02\x02"
                assert struct.unpack_from("<Q", Rx.buf, p + 10)[0] == types["children"]["00000001"]["header"]
            p += 8 + msize


def test_x_matrix_file_roundtrip(tmp_path):
    """Exec:185-192: `JLD.save(xfile, "X", X)` -- the analysed matrix as `<case>_x_matrix_<n>_<m>.jld` with the single key "X"
    (what execute(X, nkrange; save=true) writes next to the per-k results); a Float32 Matrix like the W / H of those files."""
    from nmfk_jl_amd import resultio

    X = np.asfortranarray(np.random.default_rng(3).random((7, 5)).astype(np.float32))
    fn = str(tmp_path / "nmfk_x_matrix_7_5.jld")
    resultio.save(fn, X=X)
    z = resultio.load(fn)
    assert list(z) == ["X"] and z["X"].dtype == np.float32 and z["X"].shape == (7, 5)
    np.testing.assert_array_equal(z["X"], X)
    assert not [f for f in os.listdir(tmp_path) if ".tmp" in f]  # (written to a temporary name, then renamed)


def test_x_matrix_file_of_a_sparse_input(tmp_path):
    """Round 6: execute(X, nkrange; save=true) keeps the analysed matrix next to its results also when X is sparse (Exec:185-192 saves
    `X` whatever it is).  A Julia SparseMatrixCSC is a JLD compound type this writer does not produce; the file holds the CSC's own fields under
    Julia's field names, 1-based like Julia stores them (stated deviation: DESIGN.md section 8)."""
    import scipy.sparse as sp
    from nmfk_jl_amd import resultio

    X = sp.random(9, 6, 0.4, format="csc", dtype=np.float32, random_state=2)
    fn = str(tmp_path / "nmfk_x_matrix_9_6.jld")
    resultio.save(fn, X_m=np.int64(9), X_n=np.int64(6), X_colptr=X.indptr.astype(np.int64) + 1, X_rowval=X.indices.astype(np.int64) + 1,
                  X_nzval=np.asarray(X.data))
    z = resultio.load(fn)
    assert sorted(z) == ["X_colptr", "X_m", "X_n", "X_nzval", "X_rowval"] and int(z["X_colptr"][0]) == 1 and int(z["X_colptr"][-1]) == X.nnz + 1
    Y = sp.csc_matrix((z["X_nzval"], np.asarray(z["X_rowval"]) - 1, np.asarray(z["X_colptr"]) - 1), shape=(int(z["X_m"]), int(z["X_n"])))
    assert abs(Y - X).sum() == 0
