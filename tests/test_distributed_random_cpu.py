"""Randomised protocol test of the sharded tiled driver on CPU (gloo, oracle engine): seeded world sizes 2..4, one or two tile rows
per rank, tile sizes and buffers, a partial last slab, masks that empty tiles at seams.  Expected: exactly the partition and the
segment count of the single-process tiler with white_order = parity, ids 1..N."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.metrics import adjusted_rand_index
from tests.test_distributed_cpu import OracleEngine, _free_port, synth


def make_case(seed):
    rs = np.random.RandomState(17000 + seed)
    world = int(rs.choice([2, 3, 4]))
    R = int(rs.choice([1, 2]))
    T = int(rs.choice([40, 48, 56, 64]))
    buf = int(rs.choice([6, 8, 10, 12]))
    W = int(rs.randint(90, 200))
    H = world * R * T - int(rs.randint(0, T // 2))          # the last slab may be partial (never empty)
    kw = dict(tile_size=T, buffer=buf, crown_radius=float(rs.choice([2, 3])), pixel_size=(1.0, 1.0), compactness=float(rs.choice([1.0, 10.0])))
    kind = int(rs.randint(0, 4))
    return world, R, H, W, kw, kind, int(rs.randint(0, 1 << 30))


def make_mask(H, W, kind, mseed):
    if kind == 0:
        return None
    rs = np.random.RandomState(mseed)
    yy, xx = np.mgrid[0:H, 0:W]
    if kind == 1:
        return ((yy - H / 2) ** 2 + (xx - W / 2) ** 2 < (0.5 * max(H, W)) ** 2)
    m = np.ones((H, W), bool)
    for _ in range(4 if kind == 2 else 8):   # rectangles: some swallow whole tiles on both sides of a seam
        y0, x0 = rs.randint(0, H - 10), rs.randint(0, W - 10)
        m[y0:y0 + rs.randint(8, max(9, H // 3)), x0:x0 + rs.randint(8, max(9, W // 2))] = False
    return m


def _worker(rank, world, port, H, W, R, kw, kind, mseed, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from obia_amd.distributed import ShardedTiler
        img = synth(H, W, 3, seed=mseed % 1000)
        mask = make_mask(H, W, kind, mseed)
        T = kw["tile_size"]
        lo, hi = rank * R * T, min(H, (rank + 1) * R * T)
        slab = torch.from_numpy(img[lo:hi].copy())
        mslab = None if mask is None else torch.from_numpy(mask[lo:hi].astype(np.uint8))
        ekw = dict(kw)
        t = ShardedTiler(slab, mslab, H, R, T, kw["buffer"], engine_factory=lambda im, m, Hg, row0, extra: OracleEngine(im, m, Hg, row0, ekw))
        labels, n = t.run()
        np.save(os.path.join(out, f"lab{rank}.npy"), labels.numpy())
        if rank == 0:
            np.save(os.path.join(out, "n.npy"), np.array([n]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("seed", range(int(os.environ.get("OBIA_RANDOM_SHARD_CASES", "6"))))
def test_random_sharded_case_equals_single_process(tmp_path, oracle, seed):
    from oracle import tiler
    world, R, H, W, kw, kind, mseed = make_case(seed)
    mp.spawn(_worker, args=(world, _free_port(), H, W, R, kw, kind, mseed, str(tmp_path)), nprocs=world, join=True)
    lab = np.concatenate([np.load(tmp_path / f"lab{r}.npy") for r in range(world)], 0)
    n = int(np.load(tmp_path / "n.npy")[0])
    ref, n_ref = tiler.create_tiled_segments(synth(H, W, 3, seed=mseed % 1000), make_mask(H, W, kind, mseed), white_order=1, **kw)
    assert lab.shape == ref.shape and n == n_ref, f"seed {seed}: world {world} R {R} {H}x{W} {kw} mask {kind}: n {n} vs {n_ref}"
    assert np.array_equal(lab == 0, ref == 0)
    assert adjusted_rand_index(lab, ref) == 1.0, f"seed {seed}: world {world} R {R} {H}x{W} {kw} mask {kind}"
    if n:
        ids = np.unique(lab[lab > 0])
        assert ids[0] == 1 and ids[-1] == n and len(ids) == n
