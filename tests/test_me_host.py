"""Host-side tables built by the library (no device work) against their numpy specifications in me.py."""
import numpy as np
import pytest

from cagroup3d_amd import _lib, me


def _offsets(rng, K, G, mean):
    counts = rng.poisson(mean, size=K * G).astype(np.int64)
    counts[rng.random(K * G) < 0.2] = 0                      # empty (offset, group) slots
    return np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)


@pytest.mark.parametrize("K,G,mean,maxlen", [(27, 1, 900, 256), (27, 1, 40000, 2048), (729, 18, 30, 512), (125, 18, 400, 256),
                                             (8, 1, 5000, 128), (1, 1, 7, 64), (27, 3, 2000, 4096), (27, 1, 0, 256)])
@pytest.mark.parametrize("xcd", [False, True])
def test_segment_table_of_the_library_equals_the_numpy_form(oracle, K, G, mean, maxlen, xcd):
    rng = np.random.default_rng(K * 1000 + G * 10 + int(xcd))
    with _lib.use_library(oracle):
        for trial in range(3):
            off = _offsets(rng, K, G, mean)
            if G > 1:
                rb = np.concatenate([[0], np.cumsum(rng.integers(1, 5000, size=G))]).tolist()
                n_out = rb[-1]
                rb = tuple(rb)
            else:
                rb, n_out = None, int(rng.integers(1, 200000))
            a = me._segments_numpy(off, K, G, maxlen, xcd, rb, n_out)
            b = me._segments_native(off, K, G, maxlen, xcd, rb, n_out)
            assert a.dtype == b.dtype == np.int32 and a.shape == b.shape, (a.shape, b.shape)
            assert np.array_equal(a, b)
            # a table covers every pair exactly once, whatever the order
            if a.shape[0]:
                cover = np.zeros(int(off[-1]), dtype=np.int32)
                for w, s, c in b:
                    cover[s:s + c] += 1
                    assert 0 < c <= maxlen and 0 <= w < K * G
                assert (cover == 1).all()


def test_segment_table_refuses_a_short_buffer_and_decreasing_offsets(oracle):
    import ctypes
    off = np.array([0, 600, 1200, 1800], dtype=np.int64)
    out = np.empty((2, 3), dtype=np.int32)
    n = ctypes.c_int64(0)
    f = oracle.raw("cg3d_host_segments")
    assert f(off.ctypes.data, 3, 1, 256, 0, None, 10, out.ctypes.data, 2, ctypes.cast(ctypes.pointer(n), ctypes.c_void_p)) != 0
    bad = np.array([0, 600, 500, 1800], dtype=np.int64)
    big = np.empty((64, 3), dtype=np.int32)
    assert f(bad.ctypes.data, 3, 1, 256, 0, None, 10, big.ctypes.data, 64, ctypes.cast(ctypes.pointer(n), ctypes.c_void_p)) != 0


@pytest.mark.parametrize("C", [64, 128, 512, 1024])
def test_bn_chunk_tables_of_the_library_equal_the_numpy_form(oracle, C):
    import torch
    rng = np.random.default_rng(C)
    dev = torch.device("cpu")
    with _lib.use_library(oracle):
        cases = [(0, 82107), (0, 1), (0, 0), (0, 5330)]
        for G in (2, 18):
            sizes = rng.integers(0, 4000, size=G)
            sizes[rng.random(G) < 0.2] = 0                      # empty groups
            cases.append(tuple(np.concatenate([[0], np.cumsum(sizes)]).tolist()))
        for bounds in cases:
            me._chunk_cache.clear()
            a = me._bn_chunks(tuple(bounds), dev, C, _force_numpy=True)
            me._chunk_cache.clear()
            b = me._bn_chunks(tuple(bounds), dev, C)
            me._chunk_cache.clear()
            assert a[1] == b[1] and a[5] == b[5], (bounds, a[1], b[1], a[5], b[5])
            for x, y in zip(a, b):
                if torch.is_tensor(x):
                    assert x.dtype == y.dtype and x.shape == y.shape and torch.equal(x, y), bounds
