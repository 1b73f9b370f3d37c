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


def make_mask(H, W, mode):
    """mode True / "disc": a disc; "seam": a disc with rectangles cut out that empty whole tiles on both sides of slab seams"""
    if not mode:
        return None
    yy, xx = np.mgrid[0:H, 0:W]
    m = ((yy - H / 2) ** 2 + (xx - W / 2) ** 2 < (0.48 * max(H, W)) ** 2)
    if mode == "seam":
        m = np.ones((H, W), bool)
        m[60:160, 0:55] = False          # swallows tiles above and below the first seam (row 100) in the first tile column
        m[180:215, 100:170] = False      # a band across the second seam (row 200)
        m[290:300, :] = False            # the last rows of a slab: the seam at row 300 only has segments on one side
    return m


def _worker(rank, world, port, H, W, C, R, kw, mask_on, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from obia_amd.distributed import create_tiled_segments_sharded
        img = synth(H, W, C)
        mask = make_mask(H, W, mask_on)
        T = kw["tile_size"]
        lo, hi = rank * R * T, min(H, (rank + 1) * R * T)
        slab = torch.from_numpy(img[lo:hi].copy())
        mslab = None if mask is None else torch.from_numpy(mask[lo:hi].astype(np.uint8))
        ekw = dict(kw)
        from obia_amd.distributed import ShardedTiler
        t = ShardedTiler(slab, mslab, H, R, T, kw["buffer"], engine_factory=lambda im, m, Hg, row0, extra: OracleEngine(im, m, Hg, row0, ekw))
        labels, n = t.run()
        np.save(os.path.join(out, f"lab{rank}.npy"), labels.numpy())
        np.save(os.path.join(out, f"stats{rank}.npy"), np.array([t.stats["kills_sent_up"], t.stats["kills_sent_down"],
                                                                  t.stats["foreign_ids"], t.stats["imports"]]))
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


@pytest.mark.parametrize("world,H,W,R,mask_on", [(2, 200, 230, 2, False), (2, 240, 170, 2, True), (3, 300, 150, 1, False),
                                                 (4, 400, 170, 2, "seam"), (8, 800, 130, 2, "seam")])
def test_sharded_equals_single_process_parity_order(tmp_path, oracle, world, H, W, R, mask_on):
    """world 4: two tile rows per rank, a mask that empties tiles on both sides of seams, segments dropped across seams in
    both directions (the kill lists travel up AND down) -- still exactly the single-process partition."""
    from oracle import tiler
    kw = dict(tile_size=50 if (world == 2 and not mask_on) or world >= 4 else 60 if mask_on else 100, buffer=8, crown_radius=3,
              pixel_size=(1.0, 1.0), compactness=10.0)
    T = kw["tile_size"]
    assert -(-H // T) <= world * R
    mp.spawn(_worker, args=(world, _free_port(), H, W, 3, R, kw, mask_on, str(tmp_path)), nprocs=world, join=True)
    lab = np.concatenate([np.load(tmp_path / f"lab{r}.npy") for r in range(world)], 0)
    n = int(np.load(tmp_path / "n.npy")[0])
    img = synth(H, W, 3)
    mask = make_mask(H, W, mask_on)
    ref, n_ref = tiler.create_tiled_segments(img, mask, white_order=1, **kw)
    assert lab.shape == ref.shape
    assert n == n_ref
    assert np.array_equal(lab == 0, ref == 0)
    assert adjusted_rand_index(lab, ref) == 1.0          # identical partition
    ids = np.unique(lab[lab > 0])
    assert ids[0] == 1 and ids[-1] == n and len(ids) == n
    stats = np.stack([np.load(tmp_path / f"stats{r}.npy") for r in range(world)])
    assert stats[:, 2].sum() > 0                         # foreign segments were imported
    if world == 4:
        assert stats[:, 0].sum() > 0 and stats[:, 1].sum() > 0, f"kill lists must cross seams in both directions: {stats}"


def _thread_rank(comm, img, mask, H, R, kw, out):
    from obia_amd.distributed import ShardedTiler
    T = kw["tile_size"]
    r = comm.rank
    lo, hi = r * R * T, min(H, (r + 1) * R * T)
    ekw = dict(kw)
    t = ShardedTiler(torch.from_numpy(img[lo:hi].copy()), None if mask is None else torch.from_numpy(mask[lo:hi].astype(np.uint8)),
                     H, R, T, kw["buffer"], comm=comm, engine_factory=lambda im, m, Hg, row0, extra: OracleEngine(im, m, Hg, row0, ekw))
    labels, n = t.run()
    out[r] = (labels.numpy(), n, dict(t.stats))


def test_eight_slabs_as_threads_of_one_process(oracle):
    """BASELINE configs[3]'s partition -- EIGHT slabs of two tile rows each -- with the ranks as threads of one process
    (obia_amd.distributed.ThreadComm: mailboxes instead of a process group; the protocol code is the same).  This is the form in
    which the 8-slab case also runs on the one-GPU box (tests/test_gpu_distributed.py), where 8 processes may not share the card."""
    import threading
    from obia_amd.distributed import ThreadComm
    from oracle import tiler
    world, R, H, W = 8, 2, 800, 150
    kw = dict(tile_size=50, buffer=8, crown_radius=3, pixel_size=(1.0, 1.0), compactness=10.0)
    img = synth(H, W, 3, seed=2)
    mask = np.ones((H, W), bool)
    mask[60:160, 0:55] = False; mask[180:215, 100:150] = False; mask[290:300, :] = False; mask[480:530, 30:90] = False
    comms = ThreadComm.make(world)
    out, errs = {}, []

    def run(c):
        try:
            _thread_rank(c, img, mask, H, R, kw, out)
        except BaseException as e:      # a rank that dies must not leave the others waiting for its mail for ten minutes
            errs.append((c.rank, repr(e)))
            c.abort()
            raise
    th = [threading.Thread(target=run, args=(c,), daemon=True) for c in comms]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errs, errs
    assert len(out) == world
    lab = np.concatenate([out[r][0] for r in range(world)], 0)
    ref, n_ref = tiler.create_tiled_segments(img, mask, white_order=1, **kw)
    assert all(out[r][1] == n_ref for r in range(world))
    assert np.array_equal(lab == 0, ref == 0) and adjusted_rand_index(lab, ref) == 1.0
    ids = np.unique(lab[lab > 0])
    assert ids[0] == 1 and ids[-1] == n_ref and len(ids) == n_ref
    assert sum(out[r][2]["foreign_ids"] for r in range(world)) > 0


def test_a_failing_rank_aborts_the_group_instead_of_leaving_its_neighbours_waiting(oracle):
    """an exception on ONE rank inside ShardedTiler.run() (here: its engine fails in the second white class) must take
    the communicator down from inside run() -- the neighbours are blocked waiting for that rank's seam rows -- so that every rank
    fails within seconds, not after the mailbox / process-group timeout."""
    import threading
    import time
    from obia_amd.distributed import ShardedTiler, ThreadComm
    world, R, H, W = 3, 2, 300, 120
    kw = dict(tile_size=50, buffer=8, crown_radius=3, pixel_size=(1.0, 1.0), compactness=10.0)
    img = synth(H, W, 3, seed=4)
    comms = ThreadComm.make(world)
    errs = {}

    class FailingEngine(OracleEngine):
        def run(self, white, tr_lo, tr_hi, parity=-1):
            if white and parity == 1:
                raise RuntimeError("engine failure on one rank")
            return super().run(white, tr_lo, tr_hi, parity)

    def run(c):
        T = kw["tile_size"]
        lo, hi = c.rank * R * T, min(H, (c.rank + 1) * R * T)
        eng = FailingEngine if c.rank == 1 else OracleEngine
        t = ShardedTiler(torch.from_numpy(img[lo:hi].copy()), None, H, R, T, kw["buffer"], comm=c,
                         engine_factory=lambda im, m, Hg, row0, extra: eng(im, m, Hg, row0, dict(kw)))
        try:
            t.run()
        except BaseException as e:   # (no abort() here: run() itself must have done it)
            errs[c.rank] = repr(e)
    t0 = time.time()
    th = [threading.Thread(target=run, args=(c,), daemon=True) for c in comms]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "a rank is still waiting for the failed one"
    assert set(errs) == {0, 1, 2} and "engine failure" in errs[1], errs
    assert time.time() - t0 < 60


def test_sharded_driver_rejects_small_tiles_and_unknown_kwargs():
    """tile_size <= 2 * buffer + 1 breaks the seam protocol: refused up front, like unknown SLIC kwargs."""
    import inspect
    from obia_amd import distributed
    src = inspect.getsource(distributed.ShardedTiler.__init__)
    assert "2 * self.B + 1" in src
    with pytest.raises((TypeError, NotImplementedError, AssertionError, AttributeError)):
        distributed.HipTilerEngine(torch.zeros((4, 4, 1)), None, 4, 0, 4, 1, 1.0, (1.0, 1.0), {"bogus": 1}, 0)
