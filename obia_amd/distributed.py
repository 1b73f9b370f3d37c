"""Sharded tiled SLIC: slabs of one raster on several GPUs, one process per GPU (torch.distributed over RCCL).

The reference has no multi-process mode; the semantic anchor is the single-process result of
``create_tiled_segments`` (obia/utils/tiling.py:103-291) with the white tiles taken in two parity classes of
tile rows (``white_order="parity"``): a sharded run produces the SAME label partition as a one-GPU run in that
order, because every tile sees exactly the same pixels and the same neighbouring segments.

Sharding: rank r owns ``rows_per_rank`` tile rows (a slab).  Black tiles never leave their slab: no exchange.
A white tile of a slab's first / last tile row grows ``buffer`` rows into the neighbouring slab, so:
  * once, at start : halo exchange of image (+ mask) rows, ``buffer + 1`` rows each way (send/recv over xGMI);
  * per parity class: the two tile rows that meet at a seam always have different parity, so exactly one side of
    every seam is active in a class.  Before the pass the passive side sends its ``buffer + 1`` boundary label
    rows; the active side imports them (foreign segments are registered with the pixel count it can see -- one
    that touches the outermost halo row continues beyond it and can never be "within" a window); after the pass
    the active side sends the rows back and the owner overwrites its copy;
  * at the end     : all_gather of the segment codes present in every slab -> global ids 1..N.
No all-reduce on the data path.  Segment ids travel as codes ``(owner_rank + 1) << 24 | owner_local_id``.

The tile passes themselves run in libobia_hip.so (obia_tiler_* session); this module is host plumbing.  The
compute engine is injectable so that the protocol is also exercised on CPU (gloo, world_size 2) by
tests/test_distributed_cpu.py with the oracle's tiler as engine.
"""
import ctypes
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .segmentation import make_params

CODE_SHIFT = 24                      # a segment travels as the int32 code (owner_rank + 1) << 24 | owner_local_id
ID_MASK = (1 << CODE_SHIFT) - 1
HUGE = 0xFFFFFFFF


class HipTilerEngine:
    """obia_tiler_* session of libobia_hip.so on local rows [row0, row0+H) of a (Hg, W) raster."""

    def __init__(self, img, mask, Hg, row0, tile_size, buffer, crown_radius, pixel_size, slic_kwargs, extra_ids, ctx=None):
        assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous()
        self.img, self.mask = img, mask
        H, W, C = img.shape
        self.G = torch.zeros((H, W), dtype=torch.int32, device=img.device)
        n_seg = slic_kwargs.get("n_segments", None)
        self.params = make_params(n_segments=0 if n_seg is None else n_seg, compactness=slic_kwargs.get("compactness", 10.0),
                                  max_num_iter=slic_kwargs.get("max_num_iter", 10),
                                  convert2lab=slic_kwargs.get("convert2lab", None),
                                  min_size_factor=slic_kwargs.get("min_size_factor", 0.5),
                                  max_size_factor=slic_kwargs.get("max_size_factor", 3),
                                  slic_zero=slic_kwargs.get("slic_zero", False), start_label=1, normalize_bands=True,
                                  exit_on_fixed_point=slic_kwargs.get("exit_on_fixed_point", False), sigma=slic_kwargs.get("sigma", 0), spacing=slic_kwargs.get("spacing"))
        if not slic_kwargs.get("enforce_connectivity", True):
            raise NotImplementedError("the tiled driver needs enforce_connectivity=True (segments must be connected pixel sets)")
        unknown = [k for k in slic_kwargs if k not in ("n_segments", "compactness", "max_num_iter", "convert2lab", "min_size_factor",
                                                      "max_size_factor", "slic_zero", "exit_on_fixed_point", "enforce_connectivity", "sigma", "spacing")]
        if unknown:
            raise TypeError(f"slic() got an unexpected keyword argument '{unknown[0]}'")
        self.tp = _lib.TilingParams()
        self.tp.tile_size, self.tp.buffer, self.tp.white_order = int(tile_size), int(buffer), 1
        self.tp.crown_radius, self.tp.pixel_width, self.tp.pixel_height = float(crown_radius), float(pixel_size[0]), float(pixel_size[1])
        self.ctx = ctx or _lib.default_context(img.device.index or 0)
        self.lib = _lib.load()
        torch.cuda.current_stream(img.device).synchronize()
        self.h = self.lib.obia_tiler_create(self.ctx.handle, img.data_ptr(), mask.data_ptr() if mask is not None else None,
                                            H, W, C, int(Hg), int(row0), ctypes.byref(self.tp), ctypes.byref(self.params),
                                            self.G.data_ptr(), int(extra_ids))
        if not self.h:
            raise ValueError(_lib.last_error())

    def run(self, white, tr_lo, tr_hi, parity=-1):
        torch.cuda.current_stream(self.G.device).synchronize()
        _lib.check(self.lib.obia_tiler_run(self.h, int(bool(white)), int(tr_lo), int(tr_hi), int(parity)))

    def next_id(self):
        return int(self.lib.obia_tiler_next_id(self.h))

    def set_segments(self, first_id, sizes):
        s = sizes.to(device=self.G.device, dtype=torch.int64).clamp(min=0, max=HUGE)
        s = torch.where(s >= 2 ** 31, s - 2 ** 32, s).to(torch.int32).contiguous()   # uint32 bit pattern
        torch.cuda.current_stream(self.G.device).synchronize()
        _lib.check(self.lib.obia_tiler_set_segments(self.h, int(first_id), int(s.numel()), s.data_ptr()))

    def import_seam(self, codes, my_rank, owner, fmap, code_of):
        """obia_tiler_import_seam: int32 wire codes (any shape) -> (int32 local ids of the same shape, first new id, number of new
        ids, largest owner id on the seam); fmap / code_of are updated in place"""
        c = codes.to(device=self.G.device, dtype=torch.int32).contiguous()
        out = torch.empty_like(c)
        first, n_new, t_max = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        torch.cuda.current_stream(self.G.device).synchronize()
        _lib.check(self.lib.obia_tiler_import_seam(self.h, c.data_ptr(), int(c.numel()), int(my_rank), int(owner), fmap.data_ptr(), int(fmap.numel()),
                                                   code_of.data_ptr(), out.data_ptr(), ctypes.byref(first), ctypes.byref(n_new), ctypes.byref(t_max)))
        return out, first.value, n_new.value, t_max.value

    def get_alive(self, n):
        out = torch.zeros((int(n),), dtype=torch.uint8, device=self.G.device)
        torch.cuda.current_stream(self.G.device).synchronize()
        _lib.check(self.lib.obia_tiler_get_alive(self.h, out.data_ptr(), int(n)))
        return out

    def set_alive(self, alive):
        a = alive.to(device=self.G.device, dtype=torch.uint8).contiguous()
        torch.cuda.current_stream(self.G.device).synchronize()
        _lib.check(self.lib.obia_tiler_set_alive(self.h, a.data_ptr(), int(a.numel())))

    def close(self):
        if self.h:
            self.lib.obia_tiler_destroy(self.h)
            self.h = None


class TorchComm:
    """Seam traffic over a torch.distributed process group: RCCL ("nccl", one rank per GPU, device tensors on the wire)
    or gloo (CPU tests; device tensors are staged through the host)."""

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.host_wire = dist.get_backend(group) == "gloo"

    def wire(self, t):
        """a contiguous tensor on the device the backend sends from"""
        t = t.contiguous()
        return t.cpu() if (self.host_wire and t.is_cuda) else t

    def wire_device(self, dev):
        return "cpu" if self.host_wire else dev

    def exchange(self, sends, recvs):
        """sends: [(tensor, peer)], recvs: [(buffer, peer)] -- one batch of point-to-point operations, waited for"""
        ops = [dist.P2POp(dist.isend, t, peer, self.group) for t, peer in sends]
        ops += [dist.P2POp(dist.irecv, b, peer, self.group) for b, peer in recvs]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def abort(self):
        """a rank that failed takes its communicators down, so that the neighbours blocked in a send / receive for its rows fail
        at once instead of waiting for the process-group timeout (RCCL: the abort propagates to the peers' operations; gloo: the
        peers see the closed connection).  Best effort: the exception that brought us here is what the caller sees."""
        try:
            from torch.distributed.distributed_c10d import _abort_process_group
            _abort_process_group(self.group)
        except Exception:   # noqa: BLE001  (older torch, or a backend without abort: the peers fall back on their timeout)
            pass

    def all_gather_ints(self, values, dev):
        """every rank's list of host integers (the only collective of the driver: two integers per rank, once per call)"""
        mine = torch.tensor([int(v) for v in values], dtype=torch.int64, device=self.wire_device(dev))
        allv = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allv, mine, group=self.group)
        return [[int(x) for x in v.tolist()] for v in allv]


class ThreadComm:
    """The same interface for ranks that are THREADS of one process sharing one GPU (or the CPU): mailboxes instead of a
    process group.  It exists so that the exact slab partition of a many-GPU run (BASELINE configs[3]: 8 slabs) can be
    rehearsed on a one-GPU box, where only a few processes may hold the card; the protocol code above it is the same.
    ``ThreadComm.make(world)`` returns one endpoint per rank."""

    class _Hub:
        def __init__(self, world):
            import queue
            import threading
            self.world = world
            self.box = {(s, d): queue.Queue() for s in range(world) for d in range(world)}
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world
            self.aborted = False

    def __init__(self, hub, rank):
        self.hub, self.rank, self.world = hub, rank, hub.world
        self.host_wire = False

    @staticmethod
    def make(world):
        hub = ThreadComm._Hub(world)
        return [ThreadComm(hub, r) for r in range(world)]

    def wire(self, t):
        return t.contiguous()

    def wire_device(self, dev):
        return dev

    def abort(self):
        """a rank that failed tells the others, so that nobody waits for its mail"""
        self.hub.aborted = True
        self.hub.barrier.abort()

    def _take(self, q):
        import queue
        for _ in range(1200):
            if self.hub.aborted:
                raise RuntimeError("another rank of the thread group failed")
            try:
                return q.get(timeout=0.5)
            except queue.Empty:
                continue
        raise RuntimeError("timed out waiting for a neighbour's message")

    def exchange(self, sends, recvs):
        for t, peer in sends:
            m = t.clone()                                              # the copy is what the peer reads
            if m.is_cuda:
                torch.cuda.current_stream(m.device).synchronize()
            self.hub.box[(self.rank, peer)].put(m)
        for b, peer in recvs:
            b.copy_(self._take(self.hub.box[(peer, self.rank)]).reshape(b.shape))

    def all_gather_ints(self, values, dev):
        self.hub.slots[self.rank] = [int(v) for v in values]
        self.hub.barrier.wait(timeout=600)
        out = [list(v) for v in self.hub.slots]
        self.hub.barrier.wait(timeout=600)
        return out


class _PhaseClock:
    """Host clock per phase of one sharded call (halo, black, import, class0, class1, export, relabel), opt-in (``profile=True``):
    the device is synchronised at every phase boundary, so the phases add up to the call and a slow rank's line says where it
    was slow.  Off by default: the boundaries are otherwise crossed without a device synchronisation."""

    def __init__(self, on, dev):
        self.on, self.dev, self.ms, self._t = bool(on), dev, {}, None

    def _now(self):
        if getattr(self.dev, "type", "cpu") == "cuda":
            torch.cuda.synchronize(self.dev)
        import time
        return time.perf_counter()

    def start(self):
        if self.on:
            self._t = self._now()

    def lap(self, name):
        if self.on:
            t = self._now()
            self.ms[name] = self.ms.get(name, 0.0) + (t - self._t) * 1e3
            self._t = t


class ShardedTiler:
    """One rank of the sharded driver.  ``slab`` / ``mask_slab``: this rank's rows (whole tile rows).

    Bookkeeping of foreign segments (the neighbours' segments that reach into my halo rows) is dense and lives on the
    device: ``fmap[owner]`` maps the owner's local id to my local id, ``code_of`` maps my local ids back to wire codes.
    Importing a seam therefore is a few gathers / scatters and ONE 8-byte read-back (how many new ids to reserve in the
    engine); there is no sort, search or ``unique`` on the data path and the kill lists are built and applied without
    any host round trip."""

    @staticmethod
    def halo_rows(rank, world, buffer):
        """(rows above, rows below) a rank keeps of its neighbours: buffer + 1 towards every neighbour."""
        hb = int(buffer) + 1
        return (hb if rank > 0 else 0, hb if rank < world - 1 else 0)

    def __init__(self, slab, mask_slab, global_rows, tile_rows_per_rank, tile_size, buffer, crown_radius=5,
                 pixel_size=(1.0, 1.0), engine_factory=None, group=None, ctx=None, ext_image=None, ext_mask=None, comm=None,
                 **slic_kwargs):
        self.clock = _PhaseClock(slic_kwargs.pop("profile", False), slab.device)
        self.clock.start()
        self.comm = comm if comm is not None else TorchComm(group)
        self.rank, self.world = self.comm.rank, self.comm.world
        if self.world > 127:
            raise ValueError("at most 127 ranks (segment codes keep the owner in 7 bits)")
        self.T, self.B, self.hb = int(tile_size), int(buffer), int(buffer) + 1
        if self.world > 1 and self.T <= 2 * self.B + 1:
            # the seam protocol needs a passive rank's second-to-last tile row to stay clear of the boundary rows the
            # active rank sends back, and the white windows of one parity class not to see each other
            raise ValueError(f"sharded driver: tile_size ({self.T}) must exceed 2 * buffer + 1 ({2 * self.B + 1})")
        self.R = int(tile_rows_per_rank)
        self.Hg = int(global_rows)
        self.row_lo = self.rank * self.R * self.T
        Hs, W = slab.shape[0], slab.shape[1]
        if self.row_lo + Hs > self.Hg or (Hs != self.R * self.T and self.row_lo + Hs != self.Hg):
            raise ValueError("a slab must hold whole tile rows (only the last slab may end at the raster's edge)")
        if self.world > 1 and Hs < self.hb:
            raise ValueError("slab shorter than the halo")
        self.dev = slab.device
        self.top = self.hb if self.rank > 0 else 0
        self.bot = self.hb if self.rank < self.world - 1 else 0
        # ---- halo exchange of image and mask rows (once) ------------------------------------------------------
        if mask_slab is None:
            mask_slab = torch.ones((Hs, W), dtype=torch.uint8, device=self.dev)
        mask_slab = _lib.mask_bytes(mask_slab, self.dev)
        # `slab` / `mask_slab` may be the middle rows of caller-allocated rasters that already have room for the halo rows
        # (ext_image / ext_mask, see halo_rows()): then nothing is copied.  A lone rank has no halo at all.
        He = self.top + Hs + self.bot
        if ext_image is not None:
            if tuple(ext_image.shape) != (He, W, slab.shape[2]) or ext_image.dtype != torch.float32 or not ext_image.is_contiguous():
                raise ValueError("ext_image must be a contiguous float32 (top + rows + bottom, W, C) tensor")
            ext = ext_image
        elif He == Hs and slab.dtype == torch.float32 and slab.is_contiguous():
            ext = slab
        else:
            ext = torch.empty((He, W, slab.shape[2]), dtype=torch.float32, device=self.dev)
            ext[self.top:self.top + Hs] = slab
        if ext_mask is not None:
            if tuple(ext_mask.shape) != (He, W) or ext_mask.dtype != torch.uint8 or not ext_mask.is_contiguous():
                raise ValueError("ext_mask must be a contiguous uint8 (top + rows + bottom, W) tensor")
            mext = ext_mask
        elif He == Hs and mask_slab.is_contiguous():
            mext = mask_slab
        else:
            mext = torch.empty((He, W), dtype=torch.uint8, device=self.dev)
            mext[self.top:self.top + Hs] = mask_slab
        self._exchange_rows(ext, send_top=slab[:self.hb], send_bot=slab[Hs - self.hb:])
        self._exchange_rows(mext, send_top=mask_slab[:self.hb], send_bot=mask_slab[Hs - self.hb:])
        self.Hs, self.W = Hs, W
        self.row0_ext = self.row_lo - self.top
        factory = engine_factory or (lambda img, m, Hg, row0, extra: HipTilerEngine(
            img, m, Hg, row0, tile_size, buffer, crown_radius, pixel_size, slic_kwargs, extra, ctx=ctx))
        # room for the foreign segments of three imports per seam (one per parity class it is active in, the final refresh
        # for the statistics): every imported segment has a pixel in the hb boundary rows
        self.engine = factory(ext.contiguous(), mext.contiguous(), self.Hg, self.row0_ext, 16 * W + 1024)
        self.G = self.engine.G
        gdev = self.G.device
        self.code_of = torch.zeros((1 << 16,), dtype=torch.int32, device=gdev)   # my local id -> wire code (0: one of my own)
        self.fmap = {}        # owner rank -> int32 map: the owner's local id -> my local id (0: not imported yet)
        self.f_batches = []   # (owner rank, first local id, count): contiguous ranges of imported ids
        self.stats = {"imports": 0, "foreign_ids": 0, "kills_sent_up": 0, "kills_sent_down": 0}   # host-side counters (tests)
        self._check_ids()
        self.clock.lap("halo_ms")        # halo rows of image and mask + the engine's set-up

    # ---- communication helpers ---------------------------------------------------------------------------------
    def _to_wire(self, t):
        return self.comm.wire(t)

    def _exchange_rows(self, ext, send_top, send_bot):
        """fill the halo rows of `ext` from the neighbours; they get my first / last hb rows"""
        sends, recvs, bufs = [], [], {}
        if self.rank > 0:
            bufs["up"] = torch.empty_like(self._to_wire(send_top))
            sends.append((self._to_wire(send_top), self.rank - 1))
            recvs.append((bufs["up"], self.rank - 1))
        if self.rank < self.world - 1:
            bufs["down"] = torch.empty_like(self._to_wire(send_bot))
            sends.append((self._to_wire(send_bot), self.rank + 1))
            recvs.append((bufs["down"], self.rank + 1))
        self.comm.exchange(sends, recvs)
        if "up" in bufs:
            ext[:self.top] = bufs["up"].to(ext.device)
        if "down" in bufs:
            ext[ext.shape[0] - self.bot:] = bufs["down"].to(ext.device)

    # ---- id codes --------------------------------------------------------------------------------------------------
    def _check_ids(self):
        """local ids must fit the 24-bit field of the wire codes (host integer, no device round trip)"""
        nid = int(self.engine.next_id())
        if nid >= (1 << CODE_SHIFT):
            raise RuntimeError(f"rank {self.rank}: {nid} provisional segment ids do not fit the {CODE_SHIFT}-bit id field of the "
                               "seam codes: use smaller slabs (more ranks) or larger segments")
        if nid > self.code_of.numel():
            grown = torch.zeros((max(nid, 2 * self.code_of.numel()),), dtype=torch.int32, device=self.code_of.device)
            grown[:self.code_of.numel()] = self.code_of
            self.code_of = grown
        return nid

    def _codes_of(self, ids):
        """local ids -> int32 wire codes"""
        self._check_ids()
        ids = ids.to(torch.int64)
        c = self.code_of[ids]
        own = (ids + ((self.rank + 1) << CODE_SHIFT)).to(torch.int32)
        return torch.where(c != 0, c, torch.where(ids > 0, own, torch.zeros_like(own)))

    def _ids_of(self, codes, owners):
        """int32 wire codes -> local ids; foreign codes not seen before get local ids (one contiguous range per owner,
        reserved in the engine).  ``owners``: the ranks whose segments can occur (the sender and, through it, nobody
        else: a slab is taller than a window's reach).
        The map of an owner's ids is sized to the ids that DO occur, not to the 2^24 a code can carry (round 2 kept a dense
        64-MB table per neighbour and swept it eight times per import): it starts at twice this rank's own id count -- slabs
        of one raster hold similar numbers of segments -- and doubles past the largest id on a seam; that largest id rides
        in the import's one read-back, and an import whose seam outgrew the map is simply evaluated again on the grown map."""
        if hasattr(self.engine, "import_seam") and len(owners) == 1 and not os.environ.get("OBIA_SEAM_IMPORT_TORCH"):
            return self._ids_of_kernel(codes, owners[0])
        codes = codes.to(torch.int64)
        their = codes & ID_MASK
        owner = (codes >> CODE_SHIFT) - 1
        ids = torch.where(owner == self.rank, their, torch.zeros_like(their))
        gdev = self.G.device
        for nb in owners:
            sel = (owner == nb) & (codes > 0)
            t = torch.where(sel, their, torch.zeros_like(their))            # index 0 is a dummy entry
            fm = self.fmap.get(nb)
            if fm is None:
                cap0 = 1 << max(12, (2 * int(self.engine.next_id())).bit_length())
                fm = self.fmap[nb] = torch.zeros((min(cap0, 1 << CODE_SHIFT),), dtype=torch.int32, device=gdev)
            while True:
                cap = fm.numel()
                inb = t < cap
                tc = torch.where(inb, t, torch.zeros_like(t))
                known = fm[tc] != 0
                flag = torch.zeros_like(fm)
                flag.index_put_((torch.where(sel & inb & ~known, tc, torch.zeros_like(tc)).reshape(-1),),
                                torch.ones((), dtype=torch.int32, device=gdev))
                flag[0] = 0
                rank_in_new = torch.cumsum(flag, 0, dtype=torch.int32)
                n_new, t_max = torch.stack((rank_in_new[-1].to(torch.int64), t.max())).tolist()   # the one read-back of an import
                if t_max < cap:
                    break
                grown = torch.zeros((min(1 << CODE_SHIFT, 1 << (2 * int(t_max) + 2).bit_length()),), dtype=torch.int32, device=gdev)
                grown[:cap] = fm
                fm = self.fmap[nb] = grown
                self.stats["map_growths"] = self.stats.get("map_growths", 0) + 1
            if n_new:
                first = self._check_ids()
                # reserve the range in the engine; sizes follow in _refresh_foreign_sizes
                self.engine.set_segments(first, torch.full((n_new,), HUGE, dtype=torch.int64, device=gdev))
                fm = self.fmap[nb] = torch.where(flag != 0, rank_in_new + (first - 1), fm)
                self._check_ids()
                # codes of the new ids (rank order == ascending ids of the owner); entries that are not new land on the dummy id 0
                their_all = torch.arange(fm.numel(), dtype=torch.int32, device=gdev)
                self.code_of.index_put_((torch.where(flag != 0, fm, torch.zeros_like(fm)).to(torch.int64),),
                                        torch.where(flag != 0, their_all + ((nb + 1) << CODE_SHIFT), torch.zeros_like(their_all)))
                self.code_of[0] = 0
                self.f_batches.append((nb, first, n_new))
                self.stats["foreign_ids"] += n_new
            ids = torch.where(sel, fm[tc].to(torch.int64), ids)
        self.stats["imports"] += 1
        return ids

    def _ids_of_kernel(self, codes, nb):
        """_ids_of through the library (obia_tiler_import_seam): three small kernels and one read-back behind the C ABI where the
        torch form below takes ~25 launches (Missing 3).  Same result, same bookkeeping."""
        gdev = self.G.device
        fm = self.fmap.get(nb)
        if fm is None:
            cap0 = 1 << max(12, (2 * int(self.engine.next_id())).bit_length())
            fm = self.fmap[nb] = torch.zeros((min(cap0, 1 << CODE_SHIFT),), dtype=torch.int32, device=gdev)
        while True:
            need = int(self.engine.next_id()) + int(codes.numel()) + 1          # every code could be a new segment
            if need >= (1 << CODE_SHIFT):
                need = 1 << CODE_SHIFT
            if self.code_of.numel() < need:
                grown = torch.zeros((max(need, 2 * self.code_of.numel()),), dtype=torch.int32, device=gdev)
                grown[:self.code_of.numel()] = self.code_of
                self.code_of = grown
            ids, first, n_new, t_max = self.engine.import_seam(codes, self.rank, nb, fm, self.code_of)
            if n_new:
                self._check_ids()
                self.f_batches.append((nb, first, n_new))
                self.stats["foreign_ids"] += n_new
            if t_max < fm.numel():
                break
            grown = torch.zeros((min(1 << CODE_SHIFT, 1 << (2 * int(t_max) + 2).bit_length()),), dtype=torch.int32, device=gdev)
            grown[:fm.numel()] = fm
            fm = self.fmap[nb] = grown
            self.stats["map_growths"] = self.stats.get("map_growths", 0) + 1
        self.stats["imports"] += 1
        return ids.reshape(codes.shape)      # (int32, the dtype of the label raster the callers write it into)

    def _foreign_ids(self, owner=None):
        """all local ids of imported segments (of one owner), ascending"""
        rng = [torch.arange(f, f + c, device=self.G.device) for nb, f, c in self.f_batches if owner is None or nb == owner]
        return torch.cat(rng) if rng else torch.empty((0,), dtype=torch.int64, device=self.G.device)

    def _refresh_foreign_sizes(self):
        """pixel counts of the foreign segments as seen locally; a segment on the outermost halo row continues
        beyond what this rank can see: it can never be `within` one of its windows"""
        if not self.f_batches:
            return
        G = self.G
        span = 2 * self.hb + self.B + 2
        rows, outer = [], []
        if self.top:
            rows.append(G[:span])
            outer.append(G[0])
        if self.bot:
            rows.append(G[G.shape[0] - span:])
            outer.append(G[-1])
        nmax = self._check_ids()
        counts = torch.bincount(torch.cat([r.reshape(-1) for r in rows]).to(torch.int64).clamp(min=0), minlength=nmax)[:nmax]
        counts.index_fill_(0, torch.cat(outer).to(torch.int64).clamp(min=0), HUGE)
        for nb, first, n in self.f_batches:
            self.engine.set_segments(first, counts[first:first + n])

    # ---- one parity class of white tile rows -------------------------------------------------------------------
    def _seam_roles(self, cls):
        """(active_up, passive_up, active_down, passive_down): am I the active / passive side of my upper / lower seam"""
        first_tr, last_tr = self.rank * self.R, self.rank * self.R + (-(-self.Hs // self.T)) - 1
        up = self.rank > 0
        down = self.rank < self.world - 1
        active_up = up and (first_tr & 1) == cls
        passive_up = up and not active_up              # the rank above is active on that seam (its last row has parity cls)
        active_down = down and (last_tr & 1) == cls
        passive_down = down and not active_down
        return active_up, passive_up, active_down, passive_down

    def _white_class(self, cls):
        au, pu, ad, pd = self._seam_roles(cls)
        hb, top, Hs = self.hb, self.top, self.Hs
        wdev = self.comm.wire_device(self.G.device)
        # 1. passive sides send their boundary label rows (int32 codes); active sides import them into their halo
        sends, recvs, rbuf = [], [], {}
        if pu:
            sends.append((self._to_wire(self._codes_of(self.G[top:top + hb])), self.rank - 1))
        if pd:
            sends.append((self._to_wire(self._codes_of(self.G[top + Hs - hb:top + Hs])), self.rank + 1))
        if au:
            rbuf["up"] = torch.empty((hb, self.W), dtype=torch.int32, device=wdev)
            recvs.append((rbuf["up"], self.rank - 1))
        if ad:
            rbuf["down"] = torch.empty((hb, self.W), dtype=torch.int32, device=wdev)
            recvs.append((rbuf["down"], self.rank + 1))
        self.comm.exchange(sends, recvs)
        if au:
            self.G[:top] = self._ids_of(rbuf["up"].to(self.G.device), (self.rank - 1,)).to(self.G.dtype)
        if ad:
            self.G[top + Hs:] = self._ids_of(rbuf["down"].to(self.G.device), (self.rank + 1,)).to(self.G.dtype)
        if au or ad:
            self._refresh_foreign_sizes()
        self.clock.lap("import_ms")      # boundary rows in: exchange + codes -> local ids + sizes of the imported segments
        # 2. the pass itself: every tile row of this parity in my slab
        tr_lo = self.rank * self.R
        self.engine.run(True, tr_lo, tr_lo + (-(-Hs // self.T)), cls)
        self._check_ids()
        self.clock.lap("class%d_ms" % cls)
        # 3. active sides send the halo rows back, followed by hb rows that list the neighbour's segments they dropped
        #    ([count, code, code, ...]: every imported segment has a pixel in the hb rows, so the list always fits);
        #    the owner overwrites its boundary rows and clears those segments
        sends, recvs, rbuf = [], [], {}
        if au:
            sends.append((self._to_wire(self._rows_with_kills(self.G[:top], self.rank - 1, "kills_sent_up")), self.rank - 1))
        if ad:
            sends.append((self._to_wire(self._rows_with_kills(self.G[top + Hs:], self.rank + 1, "kills_sent_down")), self.rank + 1))
        if pu:
            rbuf["up"] = torch.empty((2 * hb, self.W), dtype=torch.int32, device=wdev)
            recvs.append((rbuf["up"], self.rank - 1))
        if pd:
            rbuf["down"] = torch.empty((2 * hb, self.W), dtype=torch.int32, device=wdev)
            recvs.append((rbuf["down"], self.rank + 1))
        self.comm.exchange(sends, recvs)
        if pu:
            buf = rbuf["up"].to(self.G.device)
            self.G[top:top + hb] = self._ids_of(buf[:hb], (self.rank - 1,)).to(self.G.dtype)
            self._apply_kills(buf[hb:])
        if pd:
            buf = rbuf["down"].to(self.G.device)
            self.G[top + Hs - hb:top + Hs] = self._ids_of(buf[:hb], (self.rank + 1,)).to(self.G.dtype)
            self._apply_kills(buf[hb:])
        self.clock.lap("export_ms")      # halo rows back to their owner + the segments dropped there

    def _rows_with_kills(self, rows, owner_rank, stat):
        """int32 codes of `rows` followed by as many rows again holding [count, codes of owner_rank's segments that I dropped
        in this pass, 0, ...]; built on the device without a host round trip (compaction by prefix sum)"""
        nr = rows.shape[0]
        out = torch.zeros((2 * nr, self.W), dtype=torch.int32, device=self.G.device)
        out[:nr] = self._codes_of(rows)
        fid = self._foreign_ids(owner_rank)
        if fid.numel() > nr * self.W - 1:      # host integers: the list below holds one slot per dropped segment, after the count
            raise RuntimeError(f"rank {self.rank}: {fid.numel()} segments imported from rank {owner_rank} cannot be listed in "
                               f"{nr} x {self.W} seam entries")
        if fid.numel():
            alive = self.engine.get_alive(self._check_ids()).to(torch.bool)
            dead = ~alive[fid]
            pos = torch.cumsum(dead.to(torch.int64), 0)                     # 1-based slot of every dropped segment
            kill = out[nr:].reshape(-1)
            # every imported segment owns a pixel of the nr x W rows: the list cannot outgrow them; live entries land on slot 0,
            # which then takes the count
            kill.index_put_((torch.where(dead, pos, torch.zeros_like(pos)),), torch.where(dead, self.code_of[fid], torch.zeros_like(self.code_of[fid])))
            kill[0] = pos[-1].to(torch.int32)
            if stat in self.stats and self.G.device.type == "cpu":      # host-side counter for the CPU protocol tests only
                self.stats[stat] += int(pos[-1])
        return out

    def _apply_kills(self, rows):
        """clear the alive flag of my segments a neighbour dropped (rows: [count, code, ..., 0, ...]); no host round trip:
        unused entries are 0 and land on the dummy id 0"""
        flat = rows.reshape(-1).to(torch.int64)
        ids = flat & ID_MASK
        ids[0] = 0                                                        # the count is not an id
        alive = self.engine.get_alive(self._check_ids())
        alive.index_fill_(0, ids, 0)
        self.engine.set_alive(alive)

    def run(self):
        """all passes; returns (labels of my slab with global ids 1..N, N).  A rank that fails inside (a seam guard, a device
        error) aborts the communicator before the exception leaves: its neighbours are waiting for its rows."""
        try:
            return self._run()
        except BaseException:
            self.comm.abort()
            raise

    def _run(self):
        tr_lo = self.rank * self.R
        ntr = -(-self.Hs // self.T)
        self.clock.start()
        self.engine.run(False, tr_lo, tr_lo + ntr, -1)          # pass 1: black tiles, no communication
        self._check_ids()
        self.clock.lap("black_ms")
        self._white_class(0)
        self._white_class(1)
        labels, n = self._global_labels()
        self.clock.lap("relabel_ms")     # global ids: the one all_gather, the neighbours' numbering, the look-up
        return labels, n

    def _number_segments(self):
        """ids 1..N over all ranks: rank offsets by all_gather of the alive counts, local order = creation order.
        Returns (lut over local ids -> global id, newid of my own segments, N)."""
        nid = self._check_ids()
        alive = self.engine.get_alive(nid).to(torch.bool)
        alive[0] = False
        own = alive.clone()
        fid = self._foreign_ids()
        if fid.numel():
            own[fid] = False
        newid = torch.cumsum(own.to(torch.int64), 0) * own.to(torch.int64)        # 1-based rank among my alive segments
        n_alive = int(own.sum().item())
        allv = self.comm.all_gather_ints([n_alive, nid], self.G.device)
        counts = [v[0] for v in allv]
        nids = [v[1] for v in allv]
        offset = [sum(counts[:r]) for r in range(self.world)]
        # the neighbours' numbering of THEIR segments that live in my rows: exchange the newid tables (small)
        cdev = self.comm.wire_device(self.G.device)
        tables = {self.rank: newid}
        sends, recvs, rb = [], [], {}
        for nb in (self.rank - 1, self.rank + 1):
            if 0 <= nb < self.world:
                sends.append((self._to_wire(newid), nb))
                rb[nb] = torch.empty((nids[nb],), dtype=torch.int64, device=cdev)
                recvs.append((rb[nb], nb))
        self.comm.exchange(sends, recvs)
        for nb, t in rb.items():
            tables[nb] = t.to(self.G.device)
        lut = torch.where(own, newid + offset[self.rank], torch.zeros_like(newid))
        for nb, first, n in self.f_batches:
            their = (self.code_of[first:first + n].to(torch.int64)) & ID_MASK
            v = tables[nb][their]
            g = torch.where(v > 0, v + offset[nb], torch.zeros_like(v))
            lut[first:first + n] = torch.where(alive[first:first + n], g, torch.zeros_like(g))
        return lut.to(torch.int32), newid.to(torch.int32), sum(counts)

    def _global_labels(self):
        lut, newid, n = self._number_segments()
        self._newid = newid
        Gs = self.G[self.top:self.top + self.Hs]
        return lut[Gs.to(torch.int64)], n

    def owned_labels(self):
        """After run(): (ext_image, ext_labels, n_owned) for per-segment statistics without double counting.
        ext_* cover this rank's slab plus its halo rows; ext_labels holds dense ids 1..n_owned for the segments this
        rank OWNS (created by one of its tiles -- all their pixels lie within the halo) and 0 elsewhere, after one
        more exchange of boundary label rows so that the halo reflects the neighbours' final state."""
        top, Hs, hb = self.top, self.Hs, self.hb
        # refresh my halo label rows from the neighbours' final boundary rows (codes on the wire)
        ext_rows = torch.zeros((self.G.shape[0], self.W), dtype=torch.int32, device=self.G.device)
        send_top = self._codes_of(self.G[top:top + hb])
        send_bot = self._codes_of(self.G[top + Hs - hb:top + Hs])
        self._exchange_rows(ext_rows, send_top=send_top, send_bot=send_bot)
        if self.top:
            self.G[:top] = self._ids_of(ext_rows[:top], (self.rank - 1,)).to(self.G.dtype)
        if self.bot:
            self.G[top + Hs:] = self._ids_of(ext_rows[top + Hs:], (self.rank + 1,)).to(self.G.dtype)
        # dense ids of my own alive segments (set by run()); foreign ones -> 0: their owner counts them.  The halo
        # refresh above may have registered foreign segments that did not exist when run() numbered mine.
        nid = self._check_ids()
        newid = torch.zeros((nid,), dtype=torch.int32, device=self.G.device)
        newid[:self._newid.numel()] = self._newid
        fid = self._foreign_ids()
        if fid.numel():
            newid[fid] = 0
        dense = newid[self.G.to(torch.int64)]
        return self.engine.img if hasattr(self.engine, "img") else None, dense, int(newid.max().item()) if newid.numel() else 0

    def close(self):
        if hasattr(self.engine, "close"):
            self.engine.close()


def create_tiled_segments_sharded(slab, mask_slab=None, *, global_rows, tile_rows_per_rank, tile_size=200, buffer=30,
                                  crown_radius=5, pixel_size=(1.0, 1.0), engine_factory=None, group=None, comm=None, **slic_kwargs):
    """Sharded ``create_tiled_segments``: every rank passes its slab (whole tile rows of the global raster, rank r
    holding tile rows [r*tile_rows_per_rank, ...)).  Returns (labels of the slab with global ids 1..N, N); the
    partition equals the single-GPU result with ``white_order="parity"``."""
    t = ShardedTiler(slab, mask_slab, global_rows, tile_rows_per_rank, tile_size, buffer, crown_radius, pixel_size,
                     engine_factory, group, comm=comm, **slic_kwargs)
    try:
        return t.run()
    finally:
        t.close()
