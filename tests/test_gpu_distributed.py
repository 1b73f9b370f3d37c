"""GPU checks of the multi-GPU path on ONE card: (1) the one-GPU tiler in parity order equals the oracle's tiler
in parity order; (2) two ranks that share cuda:0 (gloo for the seam exchange, the HIP session engine for the
tile passes) reproduce the one-GPU parity-order partition exactly and own every segment exactly once."""
import os
import socket

import numpy as np
import pytest

from tests.metrics import adjusted_rand_index
from tests.test_distributed_cpu import synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_parity_order_vs_oracle(oracle):
    from obia_amd.tiling import create_tiled_segments
    from oracle import tiler
    img = synth(300, 340, 4)
    kw = dict(tile_size=100, buffer=16, crown_radius=5, pixel_size=(1.0, 1.0), compactness=10.0)
    ref, n_ref = tiler.create_tiled_segments(img, None, white_order=1, **kw)
    lab, n = create_tiled_segments(img, white_order="parity", **kw)
    assert n == n_ref and np.array_equal(lab, ref)      # same rule on both sides, bit-exact per-tile SLIC: identical rasters
    lab_r, n_r = create_tiled_segments(img, white_order="raster", **kw)
    # the two orders differ only in who wins the corner overlaps of diagonal white neighbours
    assert adjusted_rand_index(lab, lab_r) >= 0.98 and abs(n - n_r) <= 0.02 * n_r


def _worker(rank, world, port, H, W, C, R, kw, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from obia_amd.distributed import ShardedTiler
        from obia_amd.statistics import zonal_stats
        img = synth(H, W, C)
        T = kw["tile_size"]
        lo, hi = rank * R * T, min(H, (rank + 1) * R * T)
        slab = torch.from_numpy(img[lo:hi].copy()).cuda()
        t = ShardedTiler(slab, None, H, R, T, kw["buffer"], kw["crown_radius"], kw["pixel_size"], compactness=kw["compactness"])
        labels, n = t.run()
        ext_img, dense, n_owned = t.owned_labels()
        st = zonal_stats(ext_img, dense, n_labels=n_owned)
        np.save(os.path.join(out, f"lab{rank}.npy"), labels.cpu().numpy())
        np.save(os.path.join(out, f"own{rank}.npy"), np.array([n, n_owned, int(st["count"].sum().item())]))
        t.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_single_gpu(tmp_path):
    import torch.multiprocessing as mp
    from obia_amd.tiling import create_tiled_segments
    H, W, C, R = 256, 300, 4, 2
    kw = dict(tile_size=64, buffer=12, crown_radius=4, pixel_size=(1.0, 1.0), compactness=10.0)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.get_context("spawn")
    mp.spawn(_worker, args=(2, port, H, W, C, R, kw, str(tmp_path)), nprocs=2, join=True)
    lab = np.concatenate([np.load(tmp_path / f"lab{r}.npy") for r in range(2)], 0)
    ref, n_ref = create_tiled_segments(synth(H, W, C), white_order="parity", **kw)
    own = [np.load(tmp_path / f"own{r}.npy") for r in range(2)]
    assert own[0][0] == n_ref and own[0][1] + own[1][1] == n_ref          # every segment owned exactly once
    assert own[0][2] + own[1][2] == int((ref > 0).sum())                   # and every labelled pixel counted once
    assert np.array_equal(lab == 0, ref == 0)
    assert adjusted_rand_index(lab, ref) == 1.0


def _worker_masked(rank, world, port, H, W, C, R, kw, out):
    import torch
    import torch.distributed as dist
    from tests.test_distributed_cpu import make_mask
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from obia_amd.distributed import ShardedTiler
        img, mask = synth(H, W, C), make_mask(H, W, "seam")
        T = kw["tile_size"]
        lo, hi = rank * R * T, min(H, (rank + 1) * R * T)
        t = ShardedTiler(torch.from_numpy(img[lo:hi].copy()).cuda(), torch.from_numpy(mask[lo:hi].astype(np.uint8)).cuda(), H, R, T,
                         kw["buffer"], kw["crown_radius"], kw["pixel_size"], compactness=kw["compactness"])
        labels, n = t.run()
        np.save(os.path.join(out, f"lab{rank}.npy"), labels.cpu().numpy())
        np.save(os.path.join(out, f"n{rank}.npy"), np.array([n, t.stats["foreign_ids"]]))
        t.close()
    finally:
        dist.destroy_process_group()


def test_four_ranks_on_one_gpu_with_seam_mask(tmp_path):
    """Four ranks (two tile rows each) share cuda:0: the HIP session engine behind the dense foreign-id bookkeeping, a mask
    that empties tiles on both sides of seams.  Identical to the one-GPU parity-order raster."""
    import torch.multiprocessing as mp
    from obia_amd.tiling import create_tiled_segments
    from tests.test_distributed_cpu import make_mask
    H, W, C, R, world = 400, 170, 4, 2, 4
    kw = dict(tile_size=50, buffer=8, crown_radius=3, pixel_size=(1.0, 1.0), compactness=10.0)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker_masked, args=(world, port, H, W, C, R, kw, str(tmp_path)), nprocs=world, join=True)
    lab = np.concatenate([np.load(tmp_path / f"lab{r}.npy") for r in range(world)], 0)
    ref, n_ref = create_tiled_segments(synth(H, W, C), input_mask=make_mask(H, W, "seam"), white_order="parity", **kw)
    ns = [np.load(tmp_path / f"n{r}.npy") for r in range(world)]
    assert all(int(v[0]) == n_ref for v in ns) and sum(int(v[1]) for v in ns) > 0
    assert np.array_equal(lab == 0, ref == 0)
    assert adjusted_rand_index(lab, ref) == 1.0


def test_rccl_backend_wiring_with_one_rank():
    """The sharded driver over torch.distributed's nccl backend (= RCCL) with a one-rank group on this GPU: process group
    on a device, all_gather / all_reduce / barrier on device tensors, result identical to the one-GPU parity-order
    driver.  (The multi-rank exchange itself is covered by the gloo tests; it needs several GPUs to run over RCCL.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()   # a free port, not a fixed one
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "nccl_one_rank.py")], capture_output=True, text=True,
                         timeout=300, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("nccl one-rank ok")]
    assert line and line[0].endswith("True"), out.stdout[-500:]


def test_seam_import_kernel_equals_the_torch_form():
    """obia_tiler_import_seam (three kernels and one read-back behind the C ABI, VERDICT r3 Missing 3) against the torch form of
    ShardedTiler._ids_of that it replaces: the same local ids, id ranges, maps and codes over a sequence of imports -- repeats,
    codes of my own rank, codes of a third rank, an owner id beyond the map (the map is grown and the import repeated)."""
    from obia_amd.distributed import ShardedTiler, ThreadComm, CODE_SHIFT
    rs = np.random.RandomState(5)
    img = torch.as_tensor(synth(160, 200, 4)).cuda()

    def make(torch_form, monkey_env):
        comm = ThreadComm.make(1)[0]
        if torch_form:
            monkey_env["OBIA_SEAM_IMPORT_TORCH"] = "1"
        else:
            monkey_env.pop("OBIA_SEAM_IMPORT_TORCH", None)
        from obia_amd import _lib
        return ShardedTiler(img, None, 160, 2, 80, 8, 4, (1.0, 1.0), comm=comm, ctx=_lib.Context(0))   # (a session owns its context)
    a, b = make(False, os.environ), None
    try:
        seams = []
        for step in range(5):
            hi = [300, 300, 9000, 9000, 70000][step]            # (70000 lies beyond the first map: one growth)
            t = rs.randint(1, hi, size=(9, 200))
            owner = rs.choice([2, 2, 2, 1, 3], size=(9, 200))   # rank 2 is the neighbour, 1 is me, 3 somebody else
            codes = np.where(rs.rand(9, 200) < 0.2, 0, t + ((owner + 1) << CODE_SHIFT)).astype(np.int32)
            seams.append(torch.as_tensor(codes).cuda())
        a.rank = 1
        os.environ.pop("OBIA_SEAM_IMPORT_TORCH", None)
        got = [a._ids_of(c, (2,)).clone() for c in seams]
        os.environ["OBIA_SEAM_IMPORT_TORCH"] = "1"
        b = make(True, os.environ)
        b.rank = 1
        want = [b._ids_of(c, (2,)).clone() for c in seams]
        for g, w in zip(got, want):
            assert torch.equal(g.to(torch.int64), w.to(torch.int64))
        # (the same imported ids; an import that had to grow the map registers its ids in two ranges where the torch form has one)
        assert torch.equal(a._foreign_ids(), b._foreign_ids()) and a.stats["foreign_ids"] == b.stats["foreign_ids"]
        assert a.stats.get("map_growths", 0) >= 1
        n = int(a.engine.next_id())
        assert n == int(b.engine.next_id())
        assert torch.equal(a.code_of[:n], b.code_of[:n])
        assert torch.equal(a.engine.get_alive(n), b.engine.get_alive(n))
        fa, fb = a.fmap[2], b.fmap[2]
        m = min(fa.numel(), fb.numel())
        assert torch.equal(fa[:m], fb[:m]) and not fa[m:].any() and not fb[m:].any()
    finally:
        os.environ.pop("OBIA_SEAM_IMPORT_TORCH", None)
        a.close()
        if b is not None:
            b.close()
