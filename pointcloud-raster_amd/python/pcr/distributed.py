"""Row-block sharding of one grid over the GPUs of a node (one process per GPU).

Each rank owns a contiguous block of grid rows, ingests the points whose centre row falls in
its block (the engine filters by row, so a rank may be handed any superset of its points) and
accumulates glyph footprints into its block plus `halo` apron rows on each side.  The only
exchange on the data path is the neighbour halo reduce: apron rows go to the rank that owns
them and are merged with the plane's op (add for sum/weight planes, max/min otherwise) --
point-to-point send/recv over RCCL (xGMI), never a full-grid all-reduce.  The Point glyph has
no apron and therefore no exchange at all.

torch is plumbing here (device tensors viewing the engine's planes, torch.distributed for
transport); the exchange logic works on any torch tensors, so it is exercised on CPU with gloo.
"""
import torch
import torch.distributed as dist

PLANE_SUM, PLANE_WGT, PLANE_MAX, PLANE_MIN = 1, 2, 4, 8


def row_block(rank, world, height, align=1):
    """Rows [r0, r1) owned by `rank`: contiguous, balanced, block edges multiples of `align`
    (align = reference tile height makes tile-clipped glyphs exchange-free where blocks end
    on tile boundaries)."""
    units = (height + align - 1) // align
    base, extra = divmod(units, world)
    u0 = rank * base + min(rank, extra)
    u1 = u0 + base + (1 if rank < extra else 0)
    return min(u0 * align, height), min(u1 * align, height)


def _merge(dst, src, kind):
    if kind in (PLANE_SUM, PLANE_WGT):
        dst.add_(src)
    elif kind == PLANE_MAX:
        torch.maximum(dst, src, out=dst)
    elif kind == PLANE_MIN:
        torch.minimum(dst, src, out=dst)
    else:
        raise ValueError(f"unknown plane kind {kind}")


def exchange_halos(planes, own, state_row0, halo, rank, world, blocks=None, group=None):
    """Neighbour halo reduce.

    planes:     list of (tensor[state_rows, W], kind) -- this rank's state planes
    own:        (r0, r1) rows this rank owns; its planes hold rows [state_row0, state_row0 + state_rows)
    halo:       apron rows kept beyond each side of the owned block (same on every rank)
    blocks:     list of (r0, r1) of every rank (defaults to neighbours holding exactly `halo` rows)

    After the call rows [r0, r1) of every plane contain the contributions of ALL ranks.
    Apron rows are left as they are (finalize never reads them).
    A footprint never reaches further than `halo` rows, so only rank-1 and rank+1 hold data
    for this rank as long as every block is at least `halo` rows tall (checked).
    """
    if world == 1 or halo == 0 or not planes:
        return
    r0, r1 = own
    if blocks is not None:
        for b0, b1 in blocks:
            if b1 - b0 < halo and b1 > b0:
                raise ValueError("row block shorter than the glyph halo: use fewer ranks or a smaller radius")
    rows_in_state = planes[0][0].shape[0]
    s1 = state_row0 + rows_in_state
    ops, recvs = [], []
    # gloo cannot move device tensors point-to-point: stage through host memory (rehearsals and
    # tests only; the production backend is nccl = RCCL, which sends from HBM over xGMI)
    stage = dist.get_backend(group) == "gloo" and planes[0][0].is_cuda

    def outgoing(t):
        return t.cpu().contiguous() if stage else t.contiguous()

    def incoming(n, t):
        return torch.empty((n, t.shape[1]), dtype=t.dtype, device="cpu" if stage else t.device)

    for t, kind in planes:
        # apron above my block belongs to rank-1; apron below to rank+1
        up_n = r0 - state_row0          # rows I hold above my block
        dn_n = s1 - r1                  # rows I hold below my block
        if rank > 0:
            if up_n > 0:
                ops.append(dist.P2POp(dist.isend, outgoing(t[:up_n]), rank - 1, group))
            # rank-1 holds min(halo, rows I own) of my top rows
            n = min(halo, r1 - r0)
            buf = incoming(n, t)
            ops.append(dist.P2POp(dist.irecv, buf, rank - 1, group))
            recvs.append((t[r0 - state_row0: r0 - state_row0 + n], buf, kind))
        if rank < world - 1:
            if dn_n > 0:
                ops.append(dist.P2POp(dist.isend, outgoing(t[rows_in_state - dn_n:]), rank + 1, group))
            n = min(halo, r1 - r0)
            buf = incoming(n, t)
            ops.append(dist.P2POp(dist.irecv, buf, rank + 1, group))
            recvs.append((t[r1 - state_row0 - n: r1 - state_row0], buf, kind))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for dst, buf, kind in recvs:
        _merge(dst, buf.to(dst.device) if stage else buf, kind)


def allreduce_touched(touched, group=None):
    """A reference tile is 'touched' if any rank saw a valid point in it (int32 flags)."""
    if dist.get_backend(group) == "gloo" and touched.is_cuda:
        host = touched.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
        touched.copy_(host)
    else:
        dist.all_reduce(touched, op=dist.ReduceOp.MAX, group=group)


class ShardedPipeline:
    """pcr.Pipeline on this rank's row block + the halo exchange.  Usage (one process per GPU):

        sp = ShardedPipeline(cfg, rank, world)        # cfg.grid describes the WHOLE grid
        sp.ingest(cloud); ...; sp.finalize()           # finalize() = exchange + local finalize
        sp.result()                                    # rows [r0, r1) of every band
    """

    def __init__(self, cfg, rank, world, device_id=None, align=1, group=None):
        import pcr
        self.rank, self.world, self.group = rank, world, group
        self.blocks = [row_block(r, world, cfg.grid.height, align) for r in range(world)]
        self.own = self.blocks[rank]
        cfg.shard_row_begin, cfg.shard_row_end = self.own
        if device_id is not None:
            cfg.cuda_device_id = device_id
        self.pipe = pcr.Pipeline.create(cfg)
        if self.pipe is None:
            raise RuntimeError("Pipeline.create failed: " + pcr.pipeline_create_error())
        self.width = cfg.grid.width
        self.halo = self.pipe.halo_rows()
        self._views = None
        # Touched flags are per reference tile.  When every block edge falls on a tile-row boundary no tile is
        # shared between ranks and the flags are purely local: a Point-glyph run then needs no collective at all.
        th = cfg.grid.tile_height
        self.tiles_local = all(b0 % th == 0 or b0 >= cfg.grid.height for b0, _ in self.blocks)

    def _plane_tensors(self):
        if self._views is None:
            import pcr
            rows = self.pipe.state_row_count()
            self._views = []
            for ptr, kind, _group in self.pipe.state_planes():
                view = pcr.DeviceArrayView(ptr, (rows, self.width), "<f4", owner=self.pipe)
                self._views.append((torch.as_tensor(view, device="cuda"), kind))
            ptr, tx, ty = self.pipe.tile_touched_ptr()
            self._touched = torch.as_tensor(pcr.DeviceArrayView(ptr, (ty * tx,), "<i4", owner=self.pipe),
                                            device="cuda")
        return self._views

    def ingest(self, cloud):
        self.pipe.ingest(cloud)

    def exchange(self):
        """Halo reduce + touched-tile union.  No-op for a single rank."""
        # tiles_local: glyph footprints are clipped to the reference tile of their centre cell (Q4), so with
        # tile-aligned blocks nothing ever lands in a neighbour's rows -- no halo to reduce either
        if self.world == 1 or self.tiles_local:
            return
        planes = self._plane_tensors()
        # Run the collectives ON the engine's stream (wrapped as a torch ExternalStream): RCCL orders
        # itself after the scatter kernels and before the finalize kernels by stream order alone, no
        # host synchronisation inside the step.  (gloo staging copies synchronise by themselves.)
        ptr = self.pipe.stream_ptr()
        if ptr:
            ctx = torch.cuda.stream(torch.cuda.ExternalStream(ptr))
        else:                                        # pipeline on the null stream: plain host syncs
            self.pipe.synchronize()
            ctx = torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            if self.halo > 0:
                exchange_halos(planes, self.own, self.pipe.state_row_begin(), self.halo,
                               self.rank, self.world, blocks=self.blocks, group=self.group)
            if not self.tiles_local:
                allreduce_touched(self._touched, self.group)
        if not ptr:
            torch.cuda.current_stream().synchronize()

    def finalize(self):
        self.exchange()
        self.pipe.finalize()

    def result(self):
        return self.pipe.result()
