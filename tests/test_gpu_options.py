"""GPU: the programmatic options of a handle (vsc_index_set_option / _get_option: the VSC_* switches without the
environment) and caller-supplied streams (vsc_index_set_stream, vsc_tn_set_stream, vsc_set_aux_stream)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(seed=0, nq=700, nr=5000, d=128):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    r = rng.standard_normal((nr, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    r[100:110] = r[5]          # ties
    return q, r


def test_options_round_trip_and_validation(gpu):
    from vsc2022_amd import _lib
    from vsc2022_amd.vsc.index import FlatIndex

    idx = FlatIndex(64, _lib.METRIC_INNER_PRODUCT, 0)
    assert idx.get_option("prefilter") == 1 and idx.get_option("i8") == 1 and idx.get_option("i8_density") == 5e-4
    for name, value in (("i8_density", 3e-4), ("prefilter_density", 0.02), ("i8p_pair", 2), ("knn_step", 65536),
                        ("knn_ratio", 3.0), ("cand_budget", 2 ** 24), ("i8_group", 7), ("rescore_sort", 0)):
        idx.set_option(name, value)
        assert idx.get_option(name) == value, name
    with pytest.raises(ValueError):
        idx.set_option("no_such_option", 1)
    with pytest.raises(ValueError):
        idx.get_option("no_such_option")
    with pytest.raises(ValueError):
        idx.set_option("i8_density", -1.0)
    with pytest.raises(ValueError):
        idx.set_option("prefilter", 3)
    # which images of the rows are kept is decided before the first add
    idx.set_option("prefilter", 0)
    assert idx.get_option("prefilter") == 0 and idx.get_option("i8") == 0
    idx.set_option("prefilter", 2)
    idx.set_option("i8", 2)
    idx.add(np.zeros((10, 64), dtype=np.float32))
    with pytest.raises(ValueError):
        idx.set_option("prefilter", 0)
    with pytest.raises(ValueError):
        idx.set_option("i8", 0)
    with pytest.raises(ValueError):
        idx.set_option("i8_exclude", 0)
    idx.set_option("i8", 1)          # 2 -> 1 keeps the image: allowed
    idx.set_option("prefilter", 1)


@pytest.mark.parametrize("opts", [{"prefilter": 0}, {"prefilter": 2}, {"prefilter": 2, "i8": 2}, {"prefilter": 2, "i8": 0},
                                  {"prefilter": 2, "i8": 2, "i8p_pair": 2}, {"prefilter": 2, "f16_kernel": 1, "i8": 0},
                                  {"prefilter": 2, "i8": 2, "i8_sort": 0, "rescore_sort": 0},
                                  # density_hint: the caller's expected hit density picks the route of every batch
                                  # (1: exact kernel, 1e-3: fp16 pre-filter, 1e-9: int8 pre-filter)
                                  {"density_hint": 1.0}, {"density_hint": 1e-3}, {"density_hint": 1e-9}])
def test_every_route_chosen_by_option_matches_the_oracle(gpu, orc, opts):
    """the routes the test-suite forces through the environment, forced through vsc_index_set_option instead"""
    from vsc2022_amd import _lib
    from vsc2022_amd.vsc.index import FlatIndex

    q, r = _data()
    idx = FlatIndex(q.shape[1], _lib.METRIC_INNER_PRODUCT, 0)
    for k, v in opts.items():
        idx.set_option(k, v)
    idx.add(r[:3000])
    idx.add(r[3000:])
    for K in (50, 20000, 300000):
        i, j, s, rad = idx.global_topk(q, K)
        oi, oj, os_ = orc.global_threshold_search(q, r, K)
        assert np.array_equal(i, oi) and np.array_equal(j, oj) and np.array_equal(s.view(np.uint32), os_.view(np.uint32))
    D, I = idx.search(q, 7)
    oD, oI = orc.knn(q, r, 7)
    assert np.array_equal(I, oI) and np.array_equal(D.view(np.uint32), oD.view(np.uint32))


def test_handles_on_torch_streams(gpu, orc):
    """queries produced by torch kernels on a side stream, the library bound to that stream: no device synchronisation
    in between, same bits as the oracle; then back on the handle's own stream"""
    import torch
    from vsc2022_amd import _lib
    from vsc2022_amd.engine import DeviceMatcher
    from vsc2022_amd.vsc.index import FlatIndex

    q, r = _data(seed=3, nq=1200, nr=9000, d=64)
    dev = torch.device("cuda", 0)
    oi, oj, os_ = orc.global_threshold_search(q, r, 40000)
    side = torch.cuda.Stream(device=dev)
    idx = FlatIndex(q.shape[1], _lib.METRIC_INNER_PRODUCT, 0)
    big = torch.randn((4096, 4096), device=dev)
    with torch.cuda.stream(side):
        idx.use_torch_stream()
        assert idx._stream == side.cuda_stream
        rt = torch.from_numpy(r).to(dev, non_blocking=True)
        for _ in range(10):           # keep the stream busy in front of the rows the library will read
            big = big @ big
            big = big / big.abs().max()
        qt = (torch.from_numpy(q).to(dev, non_blocking=True) * 2.0) * 0.5   # produced by kernels on `side`
        idx.add(rt)
        i, j, s, rad = idx.global_topk(qt, 40000, device_out=True)
        assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(j.cpu().numpy(), oj)
        assert np.array_equal(s.cpu().numpy().view(np.uint32), os_.view(np.uint32))
    idx.use_stream(None)
    assert idx._stream is None
    i2, j2, s2, _ = idx.global_topk(qt, 40000, device_out=True)
    assert torch.equal(i, i2) and torch.equal(j, j2) and torch.equal(s, s2)
    # the engine binds index, localisation context and the handle-less entry points to torch's current stream
    off_r = np.arange(0, 9001, 30, dtype=np.int64)
    off_q = np.arange(0, 1201, 20, dtype=np.int64)
    with torch.cuda.stream(side):
        m = DeviceMatcher(rt, off_r, 0)
        m.set_queries(qt, off_q)
        res = m.match()
        assert m.index._stream == side.cuda_stream and m._tn_stream == side.cuda_stream
    m2 = DeviceMatcher(rt, off_r, 0)
    m2.set_queries(qt, off_q)
    res2 = m2.match()
    assert torch.equal(res.cand_q, res2.cand_q) and torch.equal(res.cand_r, res2.cand_r)
    assert torch.equal(res.cand_score, res2.cand_score) and torch.equal(res.boxes, res2.boxes) and torch.equal(res.nbox, res2.nbox)


def test_unsorted_hits_are_the_same_set(gpu, orc):
    """option sort_hits = 0 (the column-sharded schedule's batches): the kept hits as they lie -- the same set, the same
    radius; a list of K entries may be a truncated one"""
    from vsc2022_amd import _lib
    from vsc2022_amd.vsc.index import FlatIndex

    q, r = _data(seed=4)
    for opts in ({"prefilter": 0}, {"prefilter": 2, "i8": 2}):
        idx = FlatIndex(q.shape[1], _lib.METRIC_INNER_PRODUCT, 0)
        for k, v in opts.items():
            idx.set_option(k, v)
        idx.add(r)
        for K in (300000, 20000):
            i, j, s, rad = idx.global_topk(q, K)
            idx.set_option("sort_hits", 0)
            assert idx.get_option("sort_hits") == 0
            ui, uj, us, urad = idx.global_topk(q, K)
            idx.set_option("sort_hits", 1)
            assert urad == rad and len(us) <= K
            if len(s) < K:      # nothing was cut: the unsorted list is the whole set
                a = np.lexsort((j, i, -s.astype(np.float64)))
                b = np.lexsort((uj, ui, -us.astype(np.float64)))
                assert np.array_equal(i[a], ui[b]) and np.array_equal(j[a], uj[b])
                assert np.array_equal(s[a].view(np.uint32), us[b].view(np.uint32))
            else:               # K of more than K kept hits: every one of them lies above the radius
                assert len(us) == K and (us > rad).all()
        # seeded steady batches (what the sharded schedule calls): every pair above the radius, as a set
        rad0 = float(np.sort((q @ r.T).ravel())[-5000])
        i, j, s, _ = idx.global_topk(q, 10 ** 7, seed_radius=rad0)
        idx.set_option("sort_hits", 0)
        ui, uj, us, _ = idx.global_topk(q, 10 ** 7, seed_radius=rad0)
        a, b = np.lexsort((j, i)), np.lexsort((uj, ui))
        assert len(s) == len(us) > 0
        assert np.array_equal(i[a], ui[b]) and np.array_equal(j[a], uj[b]) and np.array_equal(s[a].view(np.uint32), us[b].view(np.uint32))


@pytest.mark.parametrize("k", [2, 7, 20, 32, 33])
def test_knn_first_tile_threshold_matches_the_oracle(gpu, orc, k):
    """exact k-NN kernel, k > 1: the per-wave k-th-largest threshold of a run's first tile (option knn_first_tile) changes
    nothing but the number of insertions -- oracle parity with it and without, ties and short reference sets included"""
    from vsc2022_amd import _lib
    from vsc2022_amd.vsc.index import FlatIndex

    q, r = _data(seed=10 + k, nq=300, nr=1500, d=96)
    r[200:260] = r[7]          # more exact ties than k in one tile
    for nr in (1500, 130, max(k, 40)):
        oD, oI = orc.knn(q, r[:nr], k)
        for ft in (1, 0):
            idx = FlatIndex(q.shape[1], _lib.METRIC_INNER_PRODUCT, 0)
            idx.set_option("prefilter", 0)
            idx.set_option("knn_first_tile", ft)
            idx.add(r[:nr])
            D, I = idx.search(q, k)
            assert np.array_equal(I, oI), (k, nr, ft)
            assert np.array_equal(D.view(np.uint32), oD.view(np.uint32)), (k, nr, ft)


def test_sort_hits_entry_point(gpu):
    """vsc_sort_hits: (score desc, row asc, ref asc) of a hit list on its own -- host and device arrays, exact ties, with and
    without the bounds that save sort passes"""
    import ctypes
    import torch
    from vsc2022_amd import _lib
    from vsc2022_amd.engine import sort_hits_device

    rng = np.random.default_rng(5)
    for n, nrow, nref in ((1, 1, 1), (1000, 17, 40), (300000, 70000, 2100000), (5000, 3, 3)):
        i = rng.integers(0, nrow, n).astype(np.int32)
        j = rng.integers(0, nref, n).astype(np.int32)
        s = rng.choice(np.float32([-1.5, 0.0, 0.25, 0.25000003, 3.0]), n) if n % 2 == 0 else rng.standard_normal(n).astype(np.float32)
        order = np.lexsort((j, i, -s.astype(np.float64)))
        for bounds in ((nrow, nref), (0, 0)):
            oi, oj, os_ = np.empty_like(i), np.empty_like(j), np.empty_like(s)
            _lib.check(_lib.lib().vsc_sort_hits(i.ctypes.data, j.ctypes.data, s.ctypes.data, n, _lib.MEM_HOST, bounds[0], bounds[1],
                                                oi.ctypes.data, oj.ctypes.data, os_.ctypes.data, _lib.MEM_HOST, 0))
            assert np.array_equal(oi, i[order]) and np.array_equal(oj, j[order])
            assert np.array_equal(os_.view(np.uint32), s[order].view(np.uint32))
        dev = torch.device("cuda", 0)
        di, dj, ds = sort_hits_device(torch.from_numpy(i).to(dev), torch.from_numpy(j).to(dev), torch.from_numpy(s).to(dev), nrow, nref)
        assert np.array_equal(di.cpu().numpy(), i[order]) and np.array_equal(dj.cpu().numpy(), j[order])
        assert np.array_equal(ds.cpu().numpy(), s[order])
