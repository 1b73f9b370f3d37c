"""CPU-side checks (no GPU, no compute calls): the C-ABI library loads and exports every symbol that
include/obia_hip.h declares, the ctypes structs match the header, and the host-side mirror of the reference
interface rejects bad arguments before touching the device."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "obia_hip.h")


@pytest.fixture(scope="module")
def lib():
    from obia_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(obia_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 15
    cdll = ctypes.CDLL(lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(cdll, s)]
    assert not missing, f"declared in include/obia_hip.h but not exported: {missing}"
    assert set(syms) == set(lib.EXPORTED_SYMBOLS), "ctypes binding table out of sync with the header"
    assert lib.load().obia_abi_version() == 1


def test_param_structs_match_header(lib):
    p = lib.SlicParams()
    lib.load().obia_slic_default_params(ctypes.byref(p))
    assert (p.n_segments, p.compactness, p.max_num_iter, p.convert2lab, p.enforce_connectivity) == (100, 10.0, 10, -1, 1)
    assert (p.min_size_factor, p.max_size_factor, p.slic_zero, p.start_label, p.normalize_bands) == (0.5, 3.0, 0, 1, 0)
    assert ctypes.sizeof(lib.SlicParams) == 3 * 8 + 8 * 4          # 3 doubles, 8 int32
    assert p.exit_on_fixed_point == 0
    assert ctypes.sizeof(lib.TilingParams) == 3 * 8 + 4 * 4


def test_no_gpu_means_loud_failure_not_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU path|no HIP device"):
        lib.Context(0)
    from obia_amd.segmentation import slic
    with pytest.raises(RuntimeError):
        slic(np.zeros((8, 8, 4), np.float32), n_segments=4)


def test_host_argument_checks_happen_before_device_use():
    from obia_amd.segmentation import slic, create_segments
    from obia_amd.tiling import create_tiled_segments
    img = np.zeros((8, 8, 4), np.float32)
    with pytest.raises(ValueError):
        slic(img, start_label=3)
    with pytest.raises(NotImplementedError):
        slic(img, sigma=2)
    with pytest.raises(NotImplementedError):
        slic(img, spacing=(1, 1, 1))
    with pytest.raises(IndexError):
        create_segments(img, segmentation_bands=[7])
    with pytest.raises(Exception, match="unknown segmentation method"):
        create_segments(img, method="felzenszwalb")
    with pytest.raises(TypeError):
        create_segments(img, bogus_kwarg=1)
    with pytest.raises(ValueError, match="only the 'slic' method"):
        create_tiled_segments(img, method="quickshift")


def test_objects_table_columns_follow_reference_order():
    from obia_amd.statistics import stats_columns
    cols = stats_columns([0, 2], [1])
    assert cols[:7] == ["segment_id", "b0_mean", "b0_variance", "b0_min", "b0_max", "b0_skewness", "b0_kurtosis"]
    assert cols[7:13] == ["b2_mean", "b2_variance", "b2_min", "b2_max", "b2_skewness", "b2_kurtosis"]
    assert cols[13:] == ["b1_contrast", "b1_dissimilarity", "b1_homogeneity", "b1_ASM", "b1_energy", "b1_correlation"]
    assert stats_columns([0], [], calc_skewness=False, calc_kurtosis=False) == ["segment_id", "b0_mean", "b0_variance", "b0_min", "b0_max"]


def test_oracle_tiler_properties(oracle):
    """The CPU restatement of the tile loops (test infrastructure): ids are 1..N, every segment is 4-connected,
    masked pixels stay 0, the crown rule sets the density."""
    from oracle import tiler
    from scipy import ndimage
    rs = np.random.RandomState(1)
    H, W = 210, 250
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.stack([400 * np.sin(xx / (11 + 3 * c)) * np.cos(yy / (13 + 2 * c)) + 1000 + rs.normal(0, 20, (H, W))
                    for c in range(4)], -1).astype(np.float32)
    mask = np.ones((H, W), bool)
    mask[:40, :60] = False
    lab, n = tiler.create_tiled_segments(img, mask, tile_size=100, buffer=16, crown_radius=5, pixel_size=(1.0, 1.0))
    ids = np.unique(lab[lab > 0])
    assert ids[0] == 1 and ids[-1] == n == len(ids)
    assert (lab[~mask] == 0).all()
    expected = mask.sum() / (np.pi * 25)
    assert 0.6 * expected <= n <= 1.4 * expected
    st = ndimage.generate_binary_structure(2, 1)
    for v in ids[::17]:
        assert ndimage.label(lab == v, st)[1] == 1
