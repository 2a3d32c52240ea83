"""Latitude-band decomposition of the global grid across the GPUs of one node.

The reference's only communication point is `swap_bounds(field, halo_size)`, an empty
stub (generic/halo_exchange_mod.f90:12-17) that a host model replaces with its own halo
exchange (UM: UM/vn10.7/sea_breeze_diag.F90:408-410).  Here each rank owns a contiguous
band of latitude rows with full longitude circles, so

* E-W periodicity is a local copy into the ghost columns,
* N-S ghost rows come from the band neighbours by point-to-point send/recv over
  torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
  tests), poles replicate their edge row (the lat clamp of the global-grid rule),
* the sigmoid's global statistics are an all-gather of 5 doubles per rank, merged in
  rank order on the device (sb_use_gathered_moments).

`BandRunner` is torch plumbing around the C ABI: device memory, streams, collectives.
The arithmetic is all in libseabreeze_hip.so.
"""
from __future__ import annotations

import numpy as np

from . import hip as _hip


def split_rows(ny: int, world: int, cost=None, min_rows: int = 1):
    """Contiguous latitude bands [(r0, r1), ...] with r1 exclusive.

    cost=None: near-equal row counts.  cost = one non-negative number per row: the rows are cut so that the bands'
    cost sums are as equal as contiguous cuts allow (SURVEY.md 8(e) "Balance": the coastal band is clustered in
    latitude, so equal row counts give unequal work).  Every band keeps at least `min_rows` rows (a band must not be
    thinner than its ghost frame).  Deterministic: every rank computes the same cuts from the same cost vector."""
    if world < 1 or ny < world * max(1, min_rows):
        raise ValueError(f"cannot cut {ny} rows into {world} bands of at least {min_rows} rows")
    if cost is None:
        base, rem = divmod(ny, world)
        out, r = [], 0
        for k in range(world):
            n = base + (1 if k < rem else 0)
            out.append((r, r + n))
            r += n
        return out
    c = np.asarray(cost, dtype=np.float64)
    if c.shape != (ny,) or np.any(c < 0):
        raise ValueError("cost must hold one non-negative number per row")
    cum = np.concatenate(([0.0], np.cumsum(c)))
    total = cum[-1]
    cuts = [0]
    for k in range(1, world):
        # the row boundary whose cumulative cost is nearest k/world of the total, inside the room the minimum
        # band height leaves on both sides
        lo, hi = cuts[-1] + min_rows, ny - (world - k) * min_rows
        target = total * k / world
        j = int(np.searchsorted(cum, target))
        if j > 0 and abs(cum[j - 1] - target) <= abs(cum[min(j, ny)] - target):
            j -= 1
        cuts.append(min(max(j, lo), hi))
    cuts.append(ny)
    return [(cuts[k], cuts[k + 1]) for k in range(world)]


def row_cost(band_mask, nz: int):
    """Cost of every latitude row in units of the algorithmic bytes of SURVEY.md 8(d): 5 per cell (the four 2-D
    inputs and sb_con) and nz + 7 per coastal-band cell (its p column, u, v and the state)."""
    band_mask = np.asarray(band_mask, dtype=bool)
    return 5.0 * band_mask.shape[1] + (nz + 7.0) * band_mask.sum(axis=1)


def fill_ew_ghosts(loc, nx: int, h: int):
    """Periodic longitude wrap into the ghost columns of a (rows, nx+2h) array (torch or numpy)."""
    if h == 0:
        return
    loc[:, :h] = loc[:, nx:nx + h]
    loc[:, nx + h:] = loc[:, h:2 * h]


def exchange_ns(loc, nyl: int, h: int, rank: int, world: int, dist, torch):
    """Fill the N-S ghost rows of a (nyl+2h, W) array from the band neighbours; pole-side
    ghost rows replicate the edge row.  `dist` is torch.distributed (any backend)."""
    if h == 0:
        return
    ops = []
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, loc[h:2 * h], rank - 1))
        ops.append(dist.P2POp(dist.irecv, loc[0:h], rank - 1))
    if rank < world - 1:
        ops.append(dist.P2POp(dist.isend, loc[nyl:nyl + h], rank + 1))
        ops.append(dist.P2POp(dist.irecv, loc[nyl + h:nyl + 2 * h], rank + 1))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    for r in reqs:
        r.wait()
    if rank == 0:
        loc[0:h] = loc[h:h + 1]
    if rank == world - 1:
        loc[nyl + h:] = loc[nyl + h - 1:nyl + h]


class BandRunner:
    """Owns one rank's device-resident fields and runs seabreeze_diag steps on them."""

    def __init__(self, ctx: _hip.Context, torch, dist, rank: int, world: int, nx: int, ny: int, nz: int,
                 halo: int, dtype=np.float64, comm: str = "torch", rows=None, static_sigma: bool = False):
        """comm="torch": ghost rows and moments travel through torch.distributed (`dist`, any backend);
        comm="native": through the library's own RCCL communicator (ctx.comm_init must have run):
        ncclSend/ncclRecv + ncclAllGather enqueued on the compute stream by two C-ABI calls.
        static_sigma=True (opt-in, sb_set_static_sigma): sigma's statistics are formed and gathered in the first
        step only; later steps run without the moments pass, the all-gather and the merge."""
        assert comm in ("torch", "native")
        self.static_sigma = bool(static_sigma)
        self._stats_done = False
        self._shared_runtime = None
        ctx.set_static_sigma(self.static_sigma)
        self.comm = comm
        self.ctx, self.torch, self.dist = ctx, torch, dist
        self.rank, self.world = rank, world
        self.nx, self.ny, self.nz = nx, ny, nz
        self.dtype = np.dtype(dtype)
        self.tdtype = torch.float64 if self.dtype == np.float64 else torch.float32
        self.r0, self.r1 = rows if rows is not None else split_rows(ny, world)[rank]   # rows: this rank's (r0, r1) of a custom cut
        self.nyl = self.r1 - self.r0
        self.h = halo if world > 1 else 0
        self.bnd = _hip.SB_BND_HALO if world > 1 else _hip.SB_BND_GLOBAL
        if world > 1 and self.nyl < self.h:
            raise ValueError(f"band of {self.nyl} rows is thinner than the halo {self.h}")
        self.dev = torch.device("cuda", torch.cuda.current_device())
        z = lambda: torch.zeros((self.nyl, nx), dtype=self.tdtype, device=self.dev)
        self.ws, self.wd, self.thc, self.sb_con = z(), z(), z(), z()
        if world > 1:
            self.mom = torch.zeros(5, dtype=torch.float64, device=self.dev)
            self.gath = torch.zeros(5 * world, dtype=torch.float64, device=self.dev)
            ctx.use_gathered_moments(self.gath.data_ptr(), world)
        ctx.set_search_radius_hint(halo)

    # -- streams -----------------------------------------------------------------------
    def _stream(self):
        """torch's current stream: torch and the library share one HIP runtime (torch imported before the library was
        loaded; hip.torch_stream_handle checks it once and refuses the other order)."""
        if self._shared_runtime is None:                # (reading the process's maps is not a per-step affair)
            _hip.torch_stream_handle(self.torch)
            self._shared_runtime = True
        return self.torch.cuda.current_stream().cuda_stream

    def synchronize(self):
        """Everything this runner enqueued has finished, whichever runtime and stream ran it."""
        self.ctx.synchronize()
        self.torch.cuda.synchronize()

    # -- uploads -----------------------------------------------------------------------
    def _halo_field(self, full2d):
        """Interior rows of this band inside a ghost-cell frame (ghosts zero until exchanged)."""
        t, h = self.torch, self.h
        loc = t.zeros((self.nyl + 2 * h, self.nx + 2 * h), dtype=self.tdtype, device=self.dev)
        src = t.from_numpy(np.ascontiguousarray(full2d[self.r0:self.r1])).to(self.dev)
        loc[h:h + self.nyl, h:h + self.nx] = src
        return loc

    def _fill_ghosts(self, loc):
        if self.world > 1:
            if self.comm == "native":
                self.ctx.swap_bounds_dev(self.dtype, loc.data_ptr(), self.nx, self.nyl, self.h, self._stream())
            else:
                exchange_ns(loc, self.nyl, self.h, self.rank, self.world, self.dist, self.torch)
                fill_ew_ghosts(loc, self.nx, self.h)

    def upload_static(self, z, sigma, mask):
        self.z, self.sigma, self.mask = (self._halo_field(a) for a in (z, sigma, mask))
        for a in (self.z, self.sigma, self.mask):      # static fields: ghosts exchanged once
            self._fill_ghosts(a)
        self.torch.cuda.synchronize()                  # (torch wrote them; the library reads them on streams of its own)
        self.ctx.synchronize()

    def upload_step_inputs(self, p, u, v, theta, local3d=False):
        """p, u, v: the full (nz, ny, nx) fields, or with local3d=True only this band's rows
        (nz, r1-r0, nx); theta: the full (ny, nx) field."""
        t = self.torch
        if local3d:
            band3 = lambda a: t.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        else:
            band3 = lambda a: t.from_numpy(np.ascontiguousarray(a[:, self.r0:self.r1])).to(self.dev)
        out = dict(p=band3(p), u=band3(u), v=band3(v), theta=self._halo_field(theta))
        assert out["p"].shape == (self.nz, self.nyl, self.nx), out["p"].shape
        t.cuda.synchronize()                           # (as above: torch wrote them, the library reads them)
        return out

    # -- one model step -------------------------------------------------------------------
    def step(self, timestep: float, tn: int, s):
        t = self.torch
        stream = self._stream()
        if self.world > 1 and self.comm == "native":
            # one C-ABI call: communication on the library's second stream under k_scan/k_wind
            self.ctx.band_seabreeze_diag_dev(self.dtype, timestep, tn, self.nx, self.nyl, self.nz, self.h,
                                             s["p"].data_ptr(), s["u"].data_ptr(), s["v"].data_ptr(),
                                             s["theta"].data_ptr(), self.mask.data_ptr(), self.z.data_ptr(),
                                             self.sigma.data_ptr(), self.ws.data_ptr(), self.wd.data_ptr(),
                                             self.thc.data_ptr(), self.sb_con.data_ptr(), stream)
            return
        if self.world > 1:
            # (torch's transport: torch and the library each bring a HIP runtime and streams of their own, and nothing
            # orders the two -- every hand-over is a synchronisation on the side that wrote last.  The native path
            # above has no such hand-over inside a step.)
            if not (self.static_sigma and self._stats_done):
                self.ctx.sigma_moments_dev(self.dtype, self.nx, self.nyl, self.h, self.sigma.data_ptr(),
                                           self.mom.data_ptr(), stream)
                if self.comm == "native":
                    self.ctx.allgather_moments_dev(self.mom.data_ptr(), self.gath.data_ptr(), stream)
                else:
                    self.ctx.synchronize()
                    self.dist.all_gather_into_tensor(self.gath, self.mom)
                self._stats_done = True
            self._fill_ghosts(s["theta"])               # theta changes every step
            if self.comm != "native":
                t.cuda.synchronize()
        self.ctx.seabreeze_diag_dev(self.dtype, timestep, tn, self.nx, self.nyl, self.nz, self.h, self.bnd,
                                    s["p"].data_ptr(), s["u"].data_ptr(), s["v"].data_ptr(),
                                    s["theta"].data_ptr(), self.mask.data_ptr(), self.z.data_ptr(),
                                    self.sigma.data_ptr(), self.ws.data_ptr(), self.wd.data_ptr(),
                                    self.thc.data_ptr(), self.sb_con.data_ptr(), stream)
