"""The sharded tiled driver (obia_amd/distributed.py) on CPU: world_size 2 and 3 over gloo, with the oracle's
tiler as the compute engine (the HIP engine needs a GPU; tests/test_gpu_distributed.py covers it).  Checks the
protocol: halo exchange, import / write-back of seam label rows, foreign-segment bookkeeping, global ids.
Expected result: exactly the partition of the single-process tiler with white_order = parity."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.metrics import adjusted_rand_index


def synth(H, W, C, seed=0):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    return np.stack([400 * np.sin(xx / (11 + 3 * c)) * np.cos(yy / (13 + 2 * c)) + 1000 + 50 * c + rs.normal(0, 20, (H, W))
                     for c in range(C)], -1).astype(np.float32)


class OracleEngine:
    """oracle.tiler.OracleTiler behind the engine interface of ShardedTiler"""

    def __init__(self, img, mask, Hg, row0, kw):
        from oracle import tiler
        self.t = tiler.OracleTiler(img.numpy(), mask.numpy(), Hg, row0, **kw)
        self.G = torch.from_numpy(self.t.G)          # shares memory with the tiler's label raster

    def run(self, white, tr_lo, tr_hi, parity=-1):
        self.t.run(white, tr_lo, tr_hi, parity)

    def next_id(self):
        return self.t.next_id

    def set_segments(self, first_id, sizes):
        self.t.set_segments(first_id, [int(v) for v in sizes.tolist()])

    def get_alive(self, n):
        a = torch.zeros((n,), dtype=torch.uint8)
        for g, v in self.t.alive.items():
            if g < n and v:
                a[g] = 1
        return a

    def set_alive(self, alive):
        for g in range(1, alive.numel()):
            if g in self.t.alive:
                self.t.alive[g] = bool(alive[g])


def _worker(rank, world, port, H, W, C, R, kw, mask_on, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from obia_amd.distributed import create_tiled_segments_sharded
        img = synth(H, W, C)
        mask = None
        if mask_on:
            yy, xx = np.mgrid[0:H, 0:W]
            mask = ((yy - H / 2) ** 2 + (xx - W / 2) ** 2 < (0.48 * max(H, W)) ** 2)
        T = kw["tile_size"]
        lo, hi = rank * R * T, min(H, (rank + 1) * R * T)
        slab = torch.from_numpy(img[lo:hi].copy())
        mslab = None if mask is None else torch.from_numpy(mask[lo:hi].astype(np.uint8))
        ekw = dict(kw)
        labels, n = create_tiled_segments_sharded(
            slab, mslab, global_rows=H, tile_rows_per_rank=R, tile_size=T, buffer=kw["buffer"],
            engine_factory=lambda im, m, Hg, row0, extra: OracleEngine(im, m, Hg, row0, ekw))
        np.save(os.path.join(out, f"lab{rank}.npy"), labels.numpy())
        if rank == 0:
            np.save(os.path.join(out, "n.npy"), np.array([n]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,H,W,R,mask_on", [(2, 200, 230, 2, False), (2, 240, 170, 2, True), (3, 300, 150, 1, False)])
def test_sharded_equals_single_process_parity_order(tmp_path, oracle, world, H, W, R, mask_on):
    from oracle import tiler
    kw = dict(tile_size=50 if world == 2 and not mask_on else 60 if mask_on else 100, buffer=8, crown_radius=3,
              pixel_size=(1.0, 1.0), compactness=10.0)
    T = kw["tile_size"]
    assert -(-H // T) <= world * R
    mp.spawn(_worker, args=(world, _free_port(), H, W, 3, R, kw, mask_on, str(tmp_path)), nprocs=world, join=True)
    lab = np.concatenate([np.load(tmp_path / f"lab{r}.npy") for r in range(world)], 0)
    n = int(np.load(tmp_path / "n.npy")[0])
    img = synth(H, W, 3)
    mask = None
    if mask_on:
        yy, xx = np.mgrid[0:H, 0:W]
        mask = ((yy - H / 2) ** 2 + (xx - W / 2) ** 2 < (0.48 * max(H, W)) ** 2)
    ref, n_ref = tiler.create_tiled_segments(img, mask, white_order=1, **kw)
    assert lab.shape == ref.shape
    assert n == n_ref
    assert np.array_equal(lab == 0, ref == 0)
    assert adjusted_rand_index(lab, ref) == 1.0          # identical partition
    ids = np.unique(lab[lab > 0])
    assert ids[0] == 1 and ids[-1] == n and len(ids) == n
