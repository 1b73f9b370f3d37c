"""Host-overhead probes of the sharded driver (obia_amd/distributed.py) on one GPU.
  python tools/shard_overhead.py plain      the sharded driver with ONE rank vs the plain tiled driver on the same slab (SIZE)
  python tools/shard_overhead.py ids        the seam import (_ids_of) at BASELINE configs[3]'s slab geometry -- 65 seam rows x 32768
                                            columns of int32 codes, ~7 000 distinct foreign segments out of ~450 000 ids of the
                                            sender -- with the id map sized to the ids that occur (round 3) and with the dense
                                            2^24-entry map of round 2
  python tools/shard_overhead.py c4         BASELINE configs[3]'s 8-slab partition as 8 threads on this one GPU: per-rank wall time of
                                            run() and of the plumbing inside it (the ranks share the card, so the figures are upper
                                            bounds of what a rank on its own GPU spends)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
DEV = "cuda" if torch.cuda.is_available() else "cpu"

if mode == "plain":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    from bench import synth_raster
    from obia_amd import _lib
    from obia_amd.tiling import create_tiled_segments
    from obia_amd.distributed import ShardedTiler
    H = W = int(os.environ.get("SIZE", 16384)); C = 8
    img = synth_raster(H, W, C, 0, torch.device("cuda"))
    mask = torch.ones((H, W), dtype=torch.uint8, device=DEV)
    ctx = _lib.Context(0)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        lab, n = create_tiled_segments(img, input_mask=mask, tile_size=2048, buffer=64, crown_radius=5, pixel_size=(0.5, 0.5), white_order="parity", ctx=ctx)
        torch.cuda.synchronize(); t1 = time.time()
        t = ShardedTiler(img, mask, H, H // 2048, 2048, 64, 5, (0.5, 0.5), ctx=ctx, compactness=10.0)
        torch.cuda.synchronize(); t2 = time.time()
        lab2, n2 = t.run()
        torch.cuda.synchronize(); t3 = time.time()
        ext, dense, no = t.owned_labels(); t.close()
        torch.cuda.synchronize(); t4 = time.time()
        print(f"plain {1e3*(t1-t0):.1f} ms | sharded: setup {1e3*(t2-t1):.1f} run {1e3*(t3-t2):.1f} owned {1e3*(t4-t3):.1f} | n {n} {n2} {no}", flush=True)
    dist.destroy_process_group()

elif mode == "ids":
    sync = torch.cuda.synchronize if DEV == "cuda" else (lambda: None)
    from obia_amd import distributed as D

    class FakeEngine:          # only what _ids_of touches
        def __init__(self, nid): self.nid = nid
        def next_id(self): return self.nid
        def set_segments(self, first, sizes): self.nid = max(self.nid, first + sizes.numel())

    def make(dense):
        t = D.ShardedTiler.__new__(D.ShardedTiler)
        t.rank, t.world = 3, 8
        t.engine = FakeEngine(450_000)
        t.G = torch.zeros((8, 8), dtype=torch.int32, device=DEV)
        t.code_of = torch.zeros((1 << 20,), dtype=torch.int32, device=DEV)
        t.fmap, t.f_batches = {}, []
        t.stats = {"imports": 0, "foreign_ids": 0}
        if dense:
            t.fmap[2] = torch.zeros((1 << D.CODE_SHIFT,), dtype=torch.int32, device=DEV)
        return t

    g = None
    W, hb = 32768, 65
    # a seam: runs of ~18 equal ids along every row, ids of four "segment rows" of the sender (its last ~7 300 ids)
    seg_of_col = (torch.arange(W, device=DEV) // 18)
    rows = torch.arange(hb, device=DEV)[:, None] // 18
    their = 440_000 + rows * (W // 18 + 1) + seg_of_col[None, :]
    codes = (their + ((2 + 1) << D.CODE_SHIFT)).to(torch.int32)
    for dense in (False, True):
        for rep in range(4):
            t = make(dense)
            sync(); t0 = time.time()
            ids = t._ids_of(codes, (2,))                 # first import: every id is new
            sync(); t1 = time.time()
            ids2 = t._ids_of(codes, (2,))                # write-back import: every id is known
            sync(); t2 = time.time()
        assert torch.equal(ids, ids2) and t.stats["foreign_ids"] == int(their.unique().numel())
        print(f"{'dense 2^24 map (round 2)' if dense else 'sized map (round 3)   '}: first import {1e3*(t1-t0):.3f} ms, repeat import {1e3*(t2-t1):.3f} ms, "
              f"map entries {t.fmap[2].numel()}, foreign ids {t.stats['foreign_ids']}", flush=True)

elif mode == "idsk":
    # the same seam through obia_tiler_import_seam (round 4: three kernels and one read-back behind the C ABI) on a real session
    from obia_amd import _lib
    from obia_amd import distributed as D
    W, hb = 32768, 65
    seg_of_col = (torch.arange(W, device=DEV) // 18)
    rows = torch.arange(hb, device=DEV)[:, None] // 18
    their = 440_000 + rows * (W // 18 + 1) + seg_of_col[None, :]
    codes = (their + ((2 + 1) << D.CODE_SHIFT)).to(torch.int32)
    img = torch.rand((256, 256, 4), device=DEV)
    for rep in range(4):
        t = D.ShardedTiler(img, None, 256, 2, 128, 16, 4, (1.0, 1.0), comm=D.ThreadComm.make(1)[0], ctx=_lib.Context(0))
        t.rank = 3
        t.engine.close()
        t.engine = D.HipTilerEngine(img, torch.ones((256, 256), dtype=torch.uint8, device=DEV), 256, 0, 128, 16, 4, (1.0, 1.0), {}, 40000, ctx=_lib.Context(0))
        t.fmap[2] = torch.zeros((1 << 20,), dtype=torch.int32, device=DEV)
        t.code_of = torch.zeros((1 << 22,), dtype=torch.int32, device=DEV)
        warm = (their - 20_000 + ((2 + 1) << D.CODE_SHIFT)).to(torch.int32)      # another 7 284 ids: the session's first import also
        t._ids_of(warm, (2,))                                                     # allocates its scratch (once per session)
        torch.cuda.synchronize(); t0 = time.time()
        ids = t._ids_of(codes, (2,))                 # every id is new
        torch.cuda.synchronize(); t1 = time.time()
        ids2 = t._ids_of(codes, (2,))                # write-back import: every id is known
        torch.cuda.synchronize(); t2 = time.time()
        assert torch.equal(ids, ids2) and t.stats["foreign_ids"] == 2 * int(their.unique().numel())
        t.close()
    print(f"obia_tiler_import_seam (round 4): import of 7 284 new ids {1e3*(t1-t0):.3f} ms, repeat import {1e3*(t2-t1):.3f} ms "
          f"(65 x 32768 codes; the torch form of round 3 below)", flush=True)

elif mode == "c4":
    import threading
    from bench import synth_raster
    from obia_amd import _lib
    from obia_amd.distributed import ShardedTiler, ThreadComm
    W = int(os.environ.get("WIDTH", 32768)); world, R, T, B, C = 8, 2, 2048, 64, 8
    H = world * R * T
    img = synth_raster(H, W, C, 0, torch.device("cuda"))
    PLUMB = ("_ids_of", "_refresh_foreign_sizes", "_rows_with_kills", "_apply_kills", "_codes_of", "_number_segments")
    res = {}

    def rank_main(comm):
        torch.cuda.set_device(0)
        ctx = _lib.Context(0)
        for rep in range(2):
            t = ShardedTiler(img[comm.rank * R * T:(comm.rank + 1) * R * T], None, H, R, T, B, 5, (0.5, 0.5), ctx=ctx, comm=comm, compactness=10.0)
            acc = {"plumbing": 0.0, "engine": 0.0, "depth": 0}
            def timed(fn, key):
                def w(*a, **k):
                    outer = acc["depth"] == 0
                    acc["depth"] += 1
                    if outer: torch.cuda.synchronize(); t0 = time.time()
                    r = fn(*a, **k)
                    if outer: torch.cuda.synchronize(); acc[key] += time.time() - t0
                    acc["depth"] -= 1
                    return r
                return w
            for name in PLUMB:
                setattr(t, name, timed(getattr(t, name), "plumbing"))
            t.engine.run = timed(t.engine.run, "engine")
            torch.cuda.synchronize(); t0 = time.time()
            lab, n = t.run()
            torch.cuda.synchronize(); dt = time.time() - t0
            t.close()
            res[(rep, comm.rank)] = (dt, acc["engine"], acc["plumbing"], n, t.stats.get("foreign_ids", 0), t.stats.get("map_growths", 0))
        ctx.close()
    th = [threading.Thread(target=rank_main, args=(c,)) for c in ThreadComm.make(world)]
    [x.start() for x in th]; [x.join() for x in th]
    for (rep, r), v in sorted(res.items()):
        print(f"rep {rep} rank {r}: run {1e3*v[0]:.1f} ms (engine passes {1e3*v[1]:.1f}, seam plumbing {1e3*v[2]:.1f}) | n {v[3]} foreign ids {v[4]} map growths {v[5]}", flush=True)
