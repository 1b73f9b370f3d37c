"""BASELINE-size runs checked through size-independent properties (no CPU oracle finishes at these sizes in
seconds): label count equal to the reference's published measurement, connectivity, minimum size, consecutive ids,
run-to-run bit-reproducibility, statistics against a NumPy recomputation on sampled segments."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def synth_gpu(H, W, C, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    yy = torch.arange(H, device="cuda", dtype=torch.float32)[:, None]
    xx = torch.arange(W, device="cuda", dtype=torch.float32)[None, :]
    out = torch.empty((H, W, C), device="cuda", dtype=torch.float32)
    for c in range(C):
        out[:, :, c] = 400.0 * torch.sin(xx / (11 + 3 * c)) * torch.cos(yy / (13 + 2 * c)) + 1000 + 50 * c \
            + 20.0 * torch.randn((H, W), device="cuda", generator=g)
    return out


def test_config2_4096x4096x4_properties():
    """BASELINE configs[1]: 4096x4096x4, n_segments=50000, compactness=10.  scikit-image 0.18.3 produced K = 51 984
    labels on this configuration: 228 x 228 grid seeds, none merged."""
    from scipy import ndimage
    from obia_amd.segmentation import slic
    from obia_amd.statistics import zonal_stats
    img = synth_gpu(4096, 4096, 4)
    lab = slic(img, n_segments=50000, compactness=10.0, _normalize_bands=True)
    lab2 = slic(img, n_segments=50000, compactness=10.0, _normalize_bands=True)
    assert torch.equal(lab, lab2)                                   # integer accumulators: bit-reproducible
    n = int(lab.max().item())
    assert n == 51984 and int(lab.min().item()) == 1
    sizes = torch.bincount(lab.reshape(-1).to(torch.int64), minlength=n + 1)[1:]
    min_size = int(0.5 * 4096 * 4096 / 51984)
    assert int(sizes.min().item()) >= min_size and int(sizes.max().item()) <= 3 * 4096 * 4096 // 51984
    l = lab.cpu().numpy()
    # every label is one 4-connected component: components of the label partition == number of labels
    structure = ndimage.generate_binary_structure(2, 1)
    sub = l[1000:1400, 2000:2500]
    for v in np.unique(sub)[::7]:
        assert ndimage.label(l[900:1500, 1900:2600] == v, structure)[1] == 1
    # raster-order numbering of first pixels
    first = np.full(n + 1, l.size, np.int64)
    np.minimum.at(first, l.ravel(), np.arange(l.size))
    assert np.all(np.diff(first[1:]) > 0)
    # statistics on sampled segments vs NumPy (float64 recomputation)
    st = zonal_stats(img, lab, n_labels=n)
    raw = img.cpu().numpy()
    for v in (1, 777, 25000, 51984):
        px = raw[l == v].astype(np.float64)
        np.testing.assert_allclose(st["mean"][v - 1].cpu().numpy(), px.mean(0), rtol=1e-6)
        np.testing.assert_allclose(st["variance"][v - 1].cpu().numpy(), px.var(0), rtol=1e-5)
        assert int(st["count"][v - 1].item()) == px.shape[0]


def test_tiled_8192_properties():
    """half-size BASELINE configs[2] (8192^2 x 8, tile 2048, overlap 64): ids 1..N, holes only where the reference
    leaves them (corner squares), whole-raster statistics add up, identical on a second run."""
    from obia_amd.tiling import create_tiled_segments
    from obia_amd.statistics import zonal_stats
    H = W = 8192
    img = synth_gpu(H, W, 8)
    kw = dict(tile_size=2048, buffer=64, crown_radius=5, pixel_size=(0.5, 0.5), compactness=10.0)
    lab, n = create_tiled_segments(img, **kw)
    lab2, n2 = create_tiled_segments(img, **kw)
    assert n == n2 and torch.equal(lab, lab2)
    assert int(lab.max().item()) == n
    present = torch.bincount(lab.reshape(-1).to(torch.int64), minlength=n + 1)
    assert int((present[1:] == 0).sum().item()) == 0               # ids 1..N all used
    holes = int(present[0].item())
    assert holes <= 8 * 64 * 64 * 2                                 # bottom corner squares of white windows at most
    expected = H * W * 0.25 / (np.pi * 25)                          # crown rule density
    assert 0.95 * expected <= n <= 1.1 * expected
    st = zonal_stats(img, lab, n_labels=n)
    assert int(st["count"].sum().item()) == H * W - holes
    tot = (st["mean"][:, 0] * st["count"].to(torch.float64)).sum().item()
    ref = img[:, :, 0].to(torch.float64)[lab > 0].sum().item()
    assert abs(tot - ref) <= 1e-9 * abs(ref)


def test_config3_full_size_16384_properties():
    """BASELINE configs[2] at FULL size -- the bench workload itself: 16384 x 16384 x 8, create_tiled_segments(tile 2048,
    overlap 64, crown radius 5, 0.5 m pixels) + zonal statistics.  No oracle finishes this in seconds; checked through
    size-independent properties: ids 1..N all present (tiling.py:289-290), holes only where the reference leaves them (the
    two bottom corner squares of the 32 white windows), the crown-rule density, every labelled pixel counted exactly once by
    the statistics, a checksum of the per-segment means against a direct sum over the raster, float64 recomputation of
    sampled segments, and bit-identical labels on a second run."""
    from obia_amd.tiling import create_tiled_segments
    from obia_amd.statistics import zonal_stats
    H = W = 16384
    img = synth_gpu(H, W, 8)
    kw = dict(tile_size=2048, buffer=64, crown_radius=5, pixel_size=(0.5, 0.5), compactness=10.0)
    mask = torch.ones((H, W), dtype=torch.uint8, device="cuda")
    lab, n = create_tiled_segments(img, input_mask=mask, **kw)
    lab2, n2 = create_tiled_segments(img, input_mask=mask, **kw)
    assert n == n2 and torch.equal(lab, lab2)
    del lab2
    assert int(lab.max().item()) == n and int(lab.min().item()) >= 0
    present = torch.bincount(lab.reshape(-1), minlength=n + 1)
    assert int((present[1:] == 0).sum().item()) == 0               # ids 1..N all used
    holes = int(present[0].item())
    assert holes <= 32 * 2 * 64 * 64                                # corner squares: buffer/2 = 32 m = 64 px a side
    expected = H * W * 0.25 / (np.pi * 25)                          # crown rule: one segment per pi * r^2 of map area
    assert 0.95 * expected <= n <= 1.1 * expected
    sizes = present[1:]
    assert int(sizes.min().item()) >= 1 and int(sizes.max().item()) <= 3 * 2176 * 2176 // 13000
    st = zonal_stats(img, lab, n_labels=n)
    assert torch.equal(st["count"], sizes.to(st["count"].dtype))    # every labelled pixel counted once, under its own id
    band = img[:, :, 3].to(torch.float64)
    tot = (st["mean"][:, 3] * st["count"].to(torch.float64)).sum().item()
    ref = band[lab > 0].sum().item()
    assert abs(tot - ref) <= 1e-9 * abs(ref)
    assert float(st["min"][:, 3].min().item()) == float(band[lab > 0].min().item())
    assert float(st["max"][:, 3].max().item()) == float(band[lab > 0].max().item())
    del band
    for v in (1, 4242, n // 2, n):                                  # sampled segments, recomputed in float64
        ys, xs = torch.nonzero(lab == v, as_tuple=True)
        px = img[ys, xs].to(torch.float64)
        assert px.shape[0] == int(st["count"][v - 1].item())
        torch.testing.assert_close(st["mean"][v - 1], px.mean(0), rtol=1e-6, atol=0)
        torch.testing.assert_close(st["variance"][v - 1], px.var(0, unbiased=False), rtol=1e-5, atol=1e-6)
        # one 4-connected component per id: flood from the first pixel inside the bounding box
        y0, y1, x0, x1 = int(ys.min()), int(ys.max()) + 1, int(xs.min()), int(xs.max()) + 1
        from scipy import ndimage
        sub = (lab[y0:y1, x0:x1] == v).cpu().numpy()
        assert ndimage.label(sub, ndimage.generate_binary_structure(2, 1))[1] == 1


# ---- BASELINE configs[3]: 32768 x 32768 x 8, tile 2048, overlap 64 -- whole on one GPU, and as its 8-slab partition ------------

@pytest.fixture(scope="module")
def raster_c4():
    """the 34-GB raster of BASELINE configs[3], generated once for the tests below (rows in 2048-row chunks)"""
    import bench
    img = bench.synth_raster(32768, 32768, 8, seed=0, device=torch.device("cuda", 0))
    yield img
    del img
    torch.cuda.empty_cache()


def _chunks(H, rows=2048):
    return [(y, min(H, y + rows)) for y in range(0, H, rows)]


def _same_partition(a, b, n):
    """Two label rasters (device, int32, ids 0..n, 0 = no segment) describe the same partition: a -> b is a function and so is
    b -> a, and the zeros coincide.  Scatter / gather per row chunk: no sort or unique over 10^9 pixels."""
    fwd = torch.full((n + 1,), -1, dtype=torch.int32, device=a.device)
    bwd = torch.full((n + 1,), -1, dtype=torch.int32, device=a.device)
    for y0, y1 in _chunks(a.shape[0]):
        ai, bi = a[y0:y1].reshape(-1).to(torch.int64), b[y0:y1].reshape(-1).to(torch.int64)
        fwd[ai] = b[y0:y1].reshape(-1)
        bwd[bi] = a[y0:y1].reshape(-1)
    ok = True
    for y0, y1 in _chunks(a.shape[0]):
        ai, bi = a[y0:y1].reshape(-1).to(torch.int64), b[y0:y1].reshape(-1).to(torch.int64)
        ok = ok and bool((fwd[ai] == b[y0:y1].reshape(-1)).all().item()) and bool((bwd[bi] == a[y0:y1].reshape(-1)).all().item())
        ok = ok and bool(((a[y0:y1] == 0) == (b[y0:y1] == 0)).all().item())
    return ok


def test_config4_full_size_32768_whole_on_one_gpu(raster_c4):
    """BASELINE configs[3] WHOLE on one MI355X (34 GB of raster, 16 x 16 tiles, ~3.3 M segments), through the properties of the
    16384^2 test above: ids 1..N all present, holes only in the corner squares of the 128 white windows, crown-rule density, every
    labelled pixel counted once by the statistics, checksum of the means against a direct sum, float64 recomputation and
    4-connectivity of sampled segments, bit-identical second run."""
    from obia_amd import _lib
    from obia_amd.tiling import create_tiled_segments
    from obia_amd.statistics import zonal_stats
    from scipy import ndimage
    img = raster_c4
    H = W = 32768
    ctx = _lib.Context(0)
    kw = dict(tile_size=2048, buffer=64, crown_radius=5, pixel_size=(0.5, 0.5), compactness=10.0, ctx=ctx)
    mask = torch.ones((H, W), dtype=torch.uint8, device="cuda")
    lab, n = create_tiled_segments(img, input_mask=mask, **kw)
    lab2, n2 = create_tiled_segments(img, input_mask=mask, **kw)
    assert n == n2 and all(torch.equal(lab[y0:y1], lab2[y0:y1]) for y0, y1 in _chunks(H))
    del lab2
    present = torch.zeros((n + 1,), dtype=torch.int64, device="cuda")
    for y0, y1 in _chunks(H):
        present += torch.bincount(lab[y0:y1].reshape(-1), minlength=n + 1)
    assert int((present[1:] == 0).sum().item()) == 0 and int(present.numel()) == n + 1      # ids 1..N all used, none above N
    holes = int(present[0].item())
    assert holes <= 128 * 2 * 64 * 64                                # corner squares: buffer/2 = 32 m = 64 px a side
    expected = H * W * 0.25 / (np.pi * 25)                          # crown rule: one segment per pi * r^2 of map area
    assert 0.95 * expected <= n <= 1.1 * expected
    sizes = present[1:]
    assert int(sizes.min().item()) >= 1 and int(sizes.max().item()) <= 3 * 2176 * 2176 // 13000
    st = zonal_stats(img, lab, n_labels=n, ctx=ctx)
    assert torch.equal(st["count"], sizes.to(st["count"].dtype))    # every labelled pixel counted once, under its own id
    ref = 0.0
    lo, hi = float("inf"), float("-inf")
    for y0, y1 in _chunks(H):
        band = img[y0:y1, :, 3].to(torch.float64)
        on = lab[y0:y1] > 0
        ref += float((band * on).sum().item())
        lo = min(lo, float(torch.where(on, band, torch.full_like(band, float("inf"))).min().item()))
        hi = max(hi, float(torch.where(on, band, torch.full_like(band, float("-inf"))).max().item()))
    tot = (st["mean"][:, 3] * st["count"].to(torch.float64)).sum().item()
    assert abs(tot - ref) <= 1e-9 * abs(ref)
    assert float(st["min"][:, 3].min().item()) == lo and float(st["max"][:, 3].max().item()) == hi
    for v in (1, 4242, n // 2, n):                                  # sampled segments, recomputed in float64
        sel = None
        for y0, y1 in _chunks(H):                                   # (a segment spans at most two chunks)
            ys, xs = torch.nonzero(lab[y0:y1] == v, as_tuple=True)
            if ys.numel():
                cur = torch.stack((ys + y0, xs), 1)
                sel = cur if sel is None else torch.cat((sel, cur))
        ys, xs = sel[:, 0], sel[:, 1]
        px = img[ys, xs].to(torch.float64)
        assert px.shape[0] == int(st["count"][v - 1].item())
        torch.testing.assert_close(st["mean"][v - 1], px.mean(0), rtol=1e-6, atol=0)
        torch.testing.assert_close(st["variance"][v - 1], px.var(0, unbiased=False), rtol=1e-5, atol=1e-6)
        y0, y1, x0, x1 = int(ys.min()), int(ys.max()) + 1, int(xs.min()), int(xs.max()) + 1
        sub = (lab[y0:y1, x0:x1] == v).cpu().numpy()
        assert ndimage.label(sub, ndimage.generate_binary_structure(2, 1))[1] == 1
    ctx.close()


def test_config4_eight_slab_partition_equals_whole_raster(raster_c4):
    """BASELINE configs[3]'s partition EXACTLY -- 8 ranks x 4096-row slabs (2 tile rows each), tile 2048, buffer 64, full width --
    on the one GPU there is: the ranks are threads of this process (ThreadComm: only a few processes may hold the card), each with
    its own obia_ctx / HIP stream and the HIP session engine; halo rows, seam label rows and kill lists travel through the same
    ShardedTiler code as over RCCL.  Result: the SAME partition as the one-GPU driver in parity order, every segment owned once,
    every labelled pixel counted once by the owners' statistics."""
    import threading
    from obia_amd import _lib
    from obia_amd.distributed import ShardedTiler, ThreadComm
    from obia_amd.statistics import zonal_stats
    from obia_amd.tiling import create_tiled_segments
    img = raster_c4
    H = W = 32768
    world, R, T, B = 8, 2, 2048, 64
    ctx0 = _lib.Context(0)
    ref, n_ref = create_tiled_segments(img, tile_size=T, buffer=B, crown_radius=5, pixel_size=(0.5, 0.5), compactness=10.0,
                                       white_order="parity", ctx=ctx0)
    ctx0.close()                                                    # its ~90-GB workspace goes back before the ranks start
    torch.cuda.empty_cache()
    lab = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    comms = ThreadComm.make(world)
    out, errs = {}, []

    def rank_main(comm):
        try:
            torch.cuda.set_device(0)
            r = comm.rank
            ctx = _lib.Context(0)
            t = ShardedTiler(img[r * R * T:(r + 1) * R * T], None, H, R, T, B, 5, (0.5, 0.5), ctx=ctx, comm=comm, compactness=10.0)
            labels, n = t.run()
            lab[r * R * T:(r + 1) * R * T] = labels
            ext_img, dense, n_owned = t.owned_labels()
            st = zonal_stats(ext_img, dense, n_labels=n_owned, ctx=ctx)
            out[r] = (n, n_owned, int(st["count"].sum().item()), dict(t.stats))
            t.close()
            ctx.close()
        except BaseException as e:
            errs.append((comm.rank, repr(e)))
            comm.abort()
            raise
    th = [threading.Thread(target=rank_main, args=(c,), daemon=True) for c in comms]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=900)
    assert not errs, errs
    assert len(out) == world
    assert all(out[r][0] == n_ref for r in range(world))
    assert sum(out[r][1] for r in range(world)) == n_ref                       # every segment owned exactly once
    labelled = sum(int((ref[y0:y1] > 0).sum().item()) for y0, y1 in _chunks(H))
    assert sum(out[r][2] for r in range(world)) == labelled                   # every labelled pixel counted once
    assert sum(out[r][3]["foreign_ids"] for r in range(world)) > 0
    assert int(lab.max().item()) == n_ref
    assert _same_partition(lab, ref, n_ref)
