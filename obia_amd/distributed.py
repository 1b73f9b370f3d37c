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

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .segmentation import make_params

CODE_SHIFT = 24
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
                                  max_size_factor=slic_kwargs.get("max_size_factor", 3), start_label=1, normalize_bands=True,
                                  exit_on_fixed_point=slic_kwargs.get("exit_on_fixed_point", False))
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


def _p2p(ops):
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


class ShardedTiler:
    """One rank of the sharded driver.  ``slab`` / ``mask_slab``: this rank's rows (whole tile rows)."""

    @staticmethod
    def halo_rows(rank, world, buffer):
        """(rows above, rows below) a rank keeps of its neighbours: buffer + 1 towards every neighbour."""
        hb = int(buffer) + 1
        return (hb if rank > 0 else 0, hb if rank < world - 1 else 0)

    def __init__(self, slab, mask_slab, global_rows, tile_rows_per_rank, tile_size, buffer, crown_radius=5,
                 pixel_size=(1.0, 1.0), engine_factory=None, group=None, ctx=None, ext_image=None, ext_mask=None,
                 **slic_kwargs):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > 127:
            raise ValueError("at most 127 ranks (segment codes keep the owner in 7 bits)")
        self.T, self.B, self.hb = int(tile_size), int(buffer), int(buffer) + 1
        self.R = int(tile_rows_per_rank)
        self.Hg = int(global_rows)
        self.row_lo = self.rank * self.R * self.T
        Hs, W = slab.shape[0], slab.shape[1]
        if self.row_lo + Hs > self.Hg or (Hs != self.R * self.T and self.row_lo + Hs != self.Hg):
            raise ValueError("a slab must hold whole tile rows (only the last slab may end at the raster's edge)")
        if self.world > 1 and Hs < self.hb:
            raise ValueError("slab shorter than the halo")
        self.dev = slab.device
        self.cpu_comm = dist.get_backend(group) == "gloo" and slab.is_cuda
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
        self.engine = factory(ext.contiguous(), mext.contiguous(), self.Hg, self.row0_ext, 16 * W + 1024)
        self.G = self.engine.G
        # foreign segments: sorted codes and their local ids; batches of contiguous local ids for re-sizing
        self.f_codes = torch.empty((0,), dtype=torch.int64, device=self.G.device)
        self.f_ids = torch.empty((0,), dtype=torch.int64, device=self.G.device)

    # ---- communication helpers ---------------------------------------------------------------------------------
    def _to_wire(self, t):
        t = t.contiguous()
        return t.cpu() if self.cpu_comm else t

    def _exchange_rows(self, ext, send_top, send_bot):
        """fill the halo rows of `ext` from the neighbours; they get my first / last hb rows"""
        ops, bufs = [], {}
        if self.rank > 0:
            bufs["up"] = torch.empty_like(self._to_wire(send_top))
            ops.append(dist.P2POp(dist.isend, self._to_wire(send_top), self.rank - 1, self.group))
            ops.append(dist.P2POp(dist.irecv, bufs["up"], self.rank - 1, self.group))
        if self.rank < self.world - 1:
            bufs["down"] = torch.empty_like(self._to_wire(send_bot))
            ops.append(dist.P2POp(dist.isend, self._to_wire(send_bot), self.rank + 1, self.group))
            ops.append(dist.P2POp(dist.irecv, bufs["down"], self.rank + 1, self.group))
        _p2p(ops)
        if "up" in bufs:
            ext[:self.top] = bufs["up"].to(ext.device)
        if "down" in bufs:
            ext[ext.shape[0] - self.bot:] = bufs["down"].to(ext.device)

    # ---- id codes --------------------------------------------------------------------------------------------------
    def _codes_of(self, ids):
        """local ids -> codes"""
        ids = ids.to(torch.int64)
        codes = torch.where(ids > 0, ids + ((self.rank + 1) << CODE_SHIFT), torch.zeros_like(ids))
        if self.f_ids.numel():
            order = torch.argsort(self.f_ids)
            sid, scode = self.f_ids[order], self.f_codes[order]
            pos = torch.searchsorted(sid, ids.reshape(-1)).clamp(max=sid.numel() - 1).reshape(ids.shape)
            hit = sid[pos] == ids
            codes = torch.where(hit, scode[pos], codes)
        return codes

    def _ids_of(self, codes):
        """codes -> local ids, registering unknown foreign codes"""
        codes = codes.to(torch.int64)
        own = (codes >> CODE_SHIFT) == (self.rank + 1)
        ids = torch.where(own, codes & ((1 << CODE_SHIFT) - 1), torch.zeros_like(codes))
        foreign = (codes > 0) & ~own
        if foreign.any():
            uniq = torch.unique(codes[foreign])
            if self.f_codes.numel():
                pos = torch.searchsorted(self.f_codes, uniq).clamp(max=self.f_codes.numel() - 1)
                known = self.f_codes[pos] == uniq
            else:
                known = torch.zeros_like(uniq, dtype=torch.bool)
            new = uniq[~known]
            if new.numel():
                first = self.engine.next_id()
                new_ids = torch.arange(first, first + new.numel(), device=codes.device, dtype=torch.int64)
                self.engine.set_segments(first, torch.full_like(new_ids, HUGE))   # reserves the ids; sizes follow
                allc = torch.cat([self.f_codes, new])
                alli = torch.cat([self.f_ids, new_ids])
                order = torch.argsort(allc)
                self.f_codes, self.f_ids = allc[order], alli[order]
            pos = torch.searchsorted(self.f_codes, codes.reshape(-1)).clamp(max=self.f_codes.numel() - 1).reshape(codes.shape)
            ids = torch.where(foreign, self.f_ids[pos], ids)
        return ids

    def _refresh_foreign_sizes(self):
        """pixel counts of the foreign segments as seen locally; a segment on the outermost halo row continues
        beyond what this rank can see: it can never be `within` one of its windows"""
        if not self.f_ids.numel():
            return
        G = self.G
        span = 2 * self.hb + self.B + 2
        rows = []
        if self.top:
            rows.append(G[:span].to(torch.int64))
        if self.bot:
            rows.append(G[G.shape[0] - span:].to(torch.int64))
        flat = torch.cat([r.reshape(-1) for r in rows])
        nmax = int(self.engine.next_id())
        counts = torch.bincount(flat.clamp(min=0), minlength=nmax)[:nmax]
        outer = []
        if self.top:
            outer.append(G[0].to(torch.int64))
        if self.bot:
            outer.append(G[-1].to(torch.int64))
        edge = torch.unique(torch.cat(outer))
        edge = edge[edge > 0]
        counts[edge] = HUGE
        ids_sorted = torch.sort(self.f_ids).values
        # contiguous runs of ids -> one set_segments call each
        brk = torch.nonzero(ids_sorted[1:] != ids_sorted[:-1] + 1).reshape(-1) + 1
        starts = [0] + brk.tolist()
        ends = brk.tolist() + [ids_sorted.numel()]
        for a, b in zip(starts, ends):
            first = int(ids_sorted[a])
            self.engine.set_segments(first, counts[first:first + (b - a)])

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
        # 1. passive sides send their boundary label rows (codes); active sides import them into their halo
        ops, rbuf = [], {}
        if pu:
            ops.append(dist.P2POp(dist.isend, self._to_wire(self._codes_of(self.G[top:top + hb])), self.rank - 1, self.group))
        if pd:
            ops.append(dist.P2POp(dist.isend, self._to_wire(self._codes_of(self.G[top + Hs - hb:top + Hs])), self.rank + 1, self.group))
        if au:
            rbuf["up"] = torch.empty((hb, self.W), dtype=torch.int64, device="cpu" if self.cpu_comm else self.G.device)
            ops.append(dist.P2POp(dist.irecv, rbuf["up"], self.rank - 1, self.group))
        if ad:
            rbuf["down"] = torch.empty((hb, self.W), dtype=torch.int64, device="cpu" if self.cpu_comm else self.G.device)
            ops.append(dist.P2POp(dist.irecv, rbuf["down"], self.rank + 1, self.group))
        _p2p(ops)
        if au:
            self.G[:top] = self._ids_of(rbuf["up"].to(self.G.device)).to(self.G.dtype)
        if ad:
            self.G[top + Hs:] = self._ids_of(rbuf["down"].to(self.G.device)).to(self.G.dtype)
        if au or ad:
            self._refresh_foreign_sizes()
        # 2. the pass itself: every tile row of this parity in my slab
        tr_lo = self.rank * self.R
        self.engine.run(True, tr_lo, tr_lo + (-(-Hs // self.T)), cls)
        # 3. active sides send the halo rows back (+ one extra row listing the neighbour's segments they dropped:
        #    [count, code, code, ...]); the owner overwrites its boundary rows and clears those segments
        ops, rbuf = [], {}
        wdev = "cpu" if self.cpu_comm else self.G.device
        if au:
            ops.append(dist.P2POp(dist.isend, self._to_wire(self._rows_with_kills(self.G[:top], self.rank - 1)), self.rank - 1, self.group))
        if ad:
            ops.append(dist.P2POp(dist.isend, self._to_wire(self._rows_with_kills(self.G[top + Hs:], self.rank + 1)), self.rank + 1, self.group))
        if pu:
            rbuf["up"] = torch.empty((hb + 1, self.W), dtype=torch.int64, device=wdev)
            ops.append(dist.P2POp(dist.irecv, rbuf["up"], self.rank - 1, self.group))
        if pd:
            rbuf["down"] = torch.empty((hb + 1, self.W), dtype=torch.int64, device=wdev)
            ops.append(dist.P2POp(dist.irecv, rbuf["down"], self.rank + 1, self.group))
        _p2p(ops)
        if pu:
            buf = rbuf["up"].to(self.G.device)
            self.G[top:top + hb] = self._ids_of(buf[:hb]).to(self.G.dtype)
            self._apply_kills(buf[hb])
        if pd:
            buf = rbuf["down"].to(self.G.device)
            self.G[top + Hs - hb:top + Hs] = self._ids_of(buf[:hb]).to(self.G.dtype)
            self._apply_kills(buf[hb])

    def _rows_with_kills(self, rows, owner_rank):
        """codes of `rows` plus one extra row [count, codes of owner_rank's segments that I dropped in this pass]"""
        out = torch.zeros((rows.shape[0] + 1, self.W), dtype=torch.int64, device=self.G.device)
        out[:rows.shape[0]] = self._codes_of(rows)
        if self.f_ids.numel():
            alive = self.engine.get_alive(self.engine.next_id()).to(torch.bool)
            mine = (self.f_codes >> CODE_SHIFT) == (owner_rank + 1)
            dead = mine & ~alive[self.f_ids]
            codes = self.f_codes[dead]
            if codes.numel() >= self.W:
                raise RuntimeError("kill list longer than a raster row")
            out[-1, 0] = codes.numel()
            out[-1, 1:1 + codes.numel()] = codes
        return out

    def _apply_kills(self, row):
        n = int(row[0].item())
        if n == 0:
            return
        ids = (row[1:1 + n] & ((1 << CODE_SHIFT) - 1)).to(torch.int64)
        alive = self.engine.get_alive(self.engine.next_id())
        alive[ids] = 0
        self.engine.set_alive(alive)

    def run(self):
        """all passes; returns (labels of my slab with global ids 1..N, N)"""
        tr_lo = self.rank * self.R
        ntr = -(-self.Hs // self.T)
        self.engine.run(False, tr_lo, tr_lo + ntr, -1)          # pass 1: black tiles, no communication
        self._white_class(0)
        self._white_class(1)
        labels, n = self._global_labels()
        return labels, n

    def _number_segments(self):
        """ids 1..N over all ranks: rank offsets by all_gather of the alive counts, local order = creation order.
        Returns (lut over local ids -> global id, newid of my own segments, N)."""
        nid = int(self.engine.next_id())
        alive = self.engine.get_alive(nid).to(torch.bool)
        alive[0] = False
        own = alive.clone()
        if self.f_ids.numel():
            own[self.f_ids] = False
        newid = torch.cumsum(own.to(torch.int64), 0) * own.to(torch.int64)        # 1-based rank among my alive segments
        n_alive = int(own.sum().item())
        cdev = "cpu" if self.cpu_comm else self.G.device
        mine = torch.tensor([n_alive, nid], dtype=torch.int64, device=cdev)
        allv = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allv, mine, group=self.group)
        counts = [int(v[0].item()) for v in allv]
        nids = [int(v[1].item()) for v in allv]
        offset = [sum(counts[:r]) for r in range(self.world)]
        # the neighbours' numbering of THEIR segments that live in my rows: exchange the newid tables (small)
        tables = {self.rank: newid}
        ops, rb = [], {}
        for nb in (self.rank - 1, self.rank + 1):
            if 0 <= nb < self.world:
                ops.append(dist.P2POp(dist.isend, self._to_wire(newid), nb, self.group))
                rb[nb] = torch.empty((nids[nb],), dtype=torch.int64, device=cdev)
                ops.append(dist.P2POp(dist.irecv, rb[nb], nb, self.group))
        _p2p(ops)
        for nb, t in rb.items():
            tables[nb] = t.to(self.G.device)
        lut = torch.where(own, newid + offset[self.rank], torch.zeros_like(newid))
        if self.f_ids.numel():
            owner = (self.f_codes >> CODE_SHIFT) - 1
            lid = self.f_codes & ((1 << CODE_SHIFT) - 1)
            g = torch.zeros_like(lid)
            for nb, t in tables.items():
                if nb == self.rank:
                    continue
                sel = owner == nb
                if sel.any():
                    v = t[lid[sel]]
                    g[sel] = torch.where(v > 0, v + offset[nb], torch.zeros_like(v))
            lut[self.f_ids] = torch.where(alive[self.f_ids], g, torch.zeros_like(g))
        return lut.to(torch.int32), newid.to(torch.int32), sum(counts)

    def _global_labels(self):
        lut, newid, n = self._number_segments()
        self._newid = newid
        Gs = self.G[self.top:self.top + self.Hs]
        assert int(Gs.max().item()) < lut.numel() and int(Gs.min().item()) >= 0
        return lut[Gs.to(torch.int64)], n

    def owned_labels(self):
        """After run(): (ext_image, ext_labels, n_owned) for per-segment statistics without double counting.
        ext_* cover this rank's slab plus its halo rows; ext_labels holds dense ids 1..n_owned for the segments this
        rank OWNS (created by one of its tiles -- all their pixels lie within the halo) and 0 elsewhere, after one
        more exchange of boundary label rows so that the halo reflects the neighbours' final state."""
        top, Hs, hb = self.top, self.Hs, self.hb
        # refresh my halo label rows from the neighbours' final boundary rows (codes on the wire)
        ext_rows = torch.zeros((self.G.shape[0], self.W), dtype=torch.int64, device=self.G.device)
        send_top = self._codes_of(self.G[top:top + hb])
        send_bot = self._codes_of(self.G[top + Hs - hb:top + Hs])
        self._exchange_rows(ext_rows, send_top=send_top, send_bot=send_bot)
        if self.top:
            self.G[:top] = self._ids_of(ext_rows[:top]).to(self.G.dtype)
        if self.bot:
            self.G[top + Hs:] = self._ids_of(ext_rows[top + Hs:]).to(self.G.dtype)
        # dense ids of my own alive segments (set by run()); foreign ones -> 0: their owner counts them.  The halo
        # refresh above may have registered foreign segments that did not exist when run() numbered mine.
        nid = int(self.engine.next_id())
        newid = torch.zeros((nid,), dtype=torch.int32, device=self.G.device)
        newid[:self._newid.numel()] = self._newid
        if self.f_ids.numel():
            newid[self.f_ids] = 0
        assert int(self.G.max().item()) < nid and int(self.G.min().item()) >= 0
        dense = newid[self.G.to(torch.int64)]
        return self.engine.img if hasattr(self.engine, "img") else None, dense, int(newid.max().item()) if newid.numel() else 0

    def close(self):
        if hasattr(self.engine, "close"):
            self.engine.close()


def create_tiled_segments_sharded(slab, mask_slab=None, *, global_rows, tile_rows_per_rank, tile_size=200, buffer=30,
                                  crown_radius=5, pixel_size=(1.0, 1.0), engine_factory=None, group=None, **slic_kwargs):
    """Sharded ``create_tiled_segments``: every rank passes its slab (whole tile rows of the global raster, rank r
    holding tile rows [r*tile_rows_per_rank, ...)).  Returns (labels of the slab with global ids 1..N, N); the
    partition equals the single-GPU result with ``white_order="parity"``."""
    t = ShardedTiler(slab, mask_slab, global_rows, tile_rows_per_rank, tile_size, buffer, crown_radius, pixel_size,
                     engine_factory, group, **slic_kwargs)
    try:
        return t.run()
    finally:
        t.close()
