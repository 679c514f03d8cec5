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


_FLT_MAX = 3.4028234663852886e38
_IDENTITY = {PLANE_SUM: 0.0, PLANE_WGT: 0.0, PLANE_MAX: -_FLT_MAX, PLANE_MIN: _FLT_MAX}


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

    After the call rows [r0, r1) of every plane contain the contributions of ALL ranks, and the apron rows that
    were sent hold the plane's identity again (0 / -FLT_MAX / FLT_MAX): their contents now live in their owner's rows,
    so a later exchange -- state survives finalize(), src/engine/pipeline.cpp:1344-1364 -- carries only what was
    accumulated since and nothing is counted twice.
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
    ops, recvs, sent = [], [], []
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
                sent.append((t[:up_n], kind))
            # rank-1 holds min(halo, rows I own) of my top rows
            n = min(halo, r1 - r0)
            buf = incoming(n, t)
            ops.append(dist.P2POp(dist.irecv, buf, rank - 1, group))
            recvs.append((t[r0 - state_row0: r0 - state_row0 + n], buf, kind))
        if rank < world - 1:
            if dn_n > 0:
                ops.append(dist.P2POp(dist.isend, outgoing(t[rows_in_state - dn_n:]), rank + 1, group))
                sent.append((t[rows_in_state - dn_n:], kind))
            n = min(halo, r1 - r0)
            buf = incoming(n, t)
            ops.append(dist.P2POp(dist.irecv, buf, rank + 1, group))
            recvs.append((t[r1 - state_row0 - n: r1 - state_row0], buf, kind))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for dst, buf, kind in recvs:
        _merge(dst, buf.to(dst.device) if stage else buf, kind)
    for rows, kind in sent:                       # (after the waits: the sends have left the rows)
        rows.fill_(_IDENTITY[kind])


def allreduce_touched(touched, group=None):
    """A reference tile is 'touched' if any rank saw a valid point in it (int32 flags)."""
    if dist.get_backend(group) == "gloo" and touched.is_cuda:
        host = touched.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
        touched.copy_(host)
    else:
        dist.all_reduce(touched, op=dist.ReduceOp.MAX, group=group)


_ELEM_BYTES = {"Float32": 4, "Int32": 4, "UInt32": 4, "Float64": 8}
_TORCH_DTYPE = {"Float32": torch.float32, "Int32": torch.int32, "UInt32": torch.int32, "Float64": torch.float64}
_TYPESTR = {"Float32": "<f4", "Int32": "<i4", "UInt32": "<i4", "Float64": "<f8"}


def _whole_grid(grid_cfg):
    """pcr_hip_grid (ctypes) of a pcr.GridConfig, every row owned: the routing decision sees the whole grid."""
    from . import _cabi as A
    b = grid_cfg.bounds
    return A.Grid(b.min_x, b.min_y, b.max_x, b.max_y, grid_cfg.cell_size_x, grid_cfg.cell_size_y,
                  grid_cfg.width, grid_cfg.height, grid_cfg.tile_width, grid_cfg.tile_height,
                  0, grid_cfg.height, 0, grid_cfg.height)


def partition_cloud(cloud, grid_cfg, blocks, stream=0):
    """Device-side partition of a device-resident cloud by row-block owner (pcr_hip_route_count /
    pcr_hip_route_scatter through the C-ABI).

    Returns (counts, arrays): counts[p] = points whose centre row lies in blocks[p] (host list of ints;
    points outside the grid are dropped, exactly the points the scatter kernels would drop), arrays =
    {name: (tensor grouped by owner, dtype name)} for "x", "y" and every channel.  Order inside a group
    is unspecified (the reductions do not depend on it: Count/Min/Max exactly, sums up to fp32 re-association,
    which the single-GPU atomics already have)."""
    import ctypes as C
    import pcr
    from . import _cabi as A
    L = A.lib()
    if cloud.location() != pcr.MemoryLocation.Device:
        raise ValueError("partition_cloud: the cloud must be device-resident (cloud.to_device())")
    n = cloud.count()
    world = len(blocks)
    if world > 64:
        raise ValueError("partition_cloud: at most 64 parts")
    splits = [blocks[0][0]] + [b1 for _, b1 in blocks]
    for (a0, a1), (b0, _) in zip(blocks, blocks[1:]):
        if a1 != b0:
            raise ValueError("partition_cloud: row blocks must be contiguous")
    names = ["x", "y"] + list(cloud.channel_names())
    kinds = {"x": "Float64", "y": "Float64"}
    for name in cloud.channel_names():
        kind = str(cloud.channel(name).dtype).split(".")[-1]
        if kind not in _ELEM_BYTES:
            raise ValueError(f"partition_cloud: channel {name!r} has dtype {kind}; only 4- and 8-byte channels are routed")
        kinds[name] = kind
    if len(names) > 8:
        raise ValueError("partition_cloud: at most 6 channels are routed in one pass")
    ptrs = cloud.device_ptrs()
    dev = torch.device("cuda", torch.cuda.current_device())
    # every torch allocation, fill and upload below runs on `stream` too (as torch's current stream), so the kernels
    # launched on it through the C-ABI are ordered after them whatever stream the caller has current
    ctx = torch.cuda.stream(torch.cuda.ExternalStream(stream)) if stream else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        if not stream:
            stream = torch.cuda.current_stream().cuda_stream
        dest = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        counts = torch.zeros(world, dtype=torch.int64, device=dev)
        g = _whole_grid(grid_cfg)
        c_splits = (C.c_int32 * (world + 1))(*splits)
        A.check(L.pcr_hip_route_count(C.byref(g), c_splits, world, ptrs["x"], ptrs["y"], None, n,
                                      dest.data_ptr(), counts.data_ptr(), stream))
        h_counts = [int(c) for c in counts.cpu().tolist()]            # D2H on the same stream: synchronizes it
        total = sum(h_counts)
        cursors = torch.tensor([sum(h_counts[:p]) for p in range(world)], dtype=torch.int64).to(dev, non_blocking=False)
        grouped = {}
        srcs, dsts, elems = [], [], []
        for name in names:
            t = torch.empty(max(total, 1), dtype=_TORCH_DTYPE[kinds[name]], device=dev)
            grouped[name] = (t[:total], kinds[name])
            srcs.append(ptrs[name])
            dsts.append(t.data_ptr())
            elems.append(_ELEM_BYTES[kinds[name]])
        k = len(names)
        A.check(L.pcr_hip_route_scatter(dest.data_ptr(), n, world, cursors.data_ptr(), k,
                                        (C.c_void_p * k)(*srcs), (C.c_void_p * k)(*dsts), (C.c_int32 * k)(*elems), stream))
        # dest / counts / cursors are released to torch's allocator when this frame ends: safe, because the allocator
        # reuses a block only on the stream it was allocated on -- this one -- i.e. after the kernels above
    return h_counts, grouped


def route_cloud(cloud, grid_cfg, blocks, rank, world, group=None, stream=0, comm=None):
    """Routes an arbitrary shard of the cloud to the owners of the row blocks: device-side partition, then one
    all-to-all per array (RCCL over xGMI; gloo stages through host memory in rehearsals).  Collective: every rank
    calls it.  Returns a device-resident pcr.PointCloud holding exactly the points whose centre row this rank owns.
    comm: a pcr_hip_comm handle -- the groups then travel through the library's own pcr_hip_comm_alltoallv (counts agreed
    first, x, y and every channel in ONE grouped ncclSend / ncclRecv round) instead of torch.distributed."""
    import pcr
    # one stream for the partition kernels, the torch allocations and the collectives (see partition_cloud)
    ctx = torch.cuda.stream(torch.cuda.ExternalStream(stream)) if stream else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        return _route_cloud_on_current_stream(cloud, grid_cfg, blocks, rank, world, group,
                                              stream or torch.cuda.current_stream().cuda_stream, comm)


def _alltoallv_native(comm, cloud, counts, grouped, world, stream):
    """pcr_hip_comm_alltoall_counts + pcr_hip_comm_alltoallv (include/pcr_hip.h) on the partition's grouped arrays."""
    import ctypes as C
    import pcr
    from . import _cabi as A
    L = A.lib()
    send = (C.c_uint64 * 64)(*counts)
    recv = (C.c_uint64 * 64)()
    A.check(L.pcr_hip_comm_alltoall_counts(comm, send, recv, stream))
    total = sum(int(recv[p]) for p in range(world))
    out = pcr.PointCloud.create(max(total, 1), pcr.MemoryLocation.Device)
    if out is not None:
        for name in cloud.channel_names():
            out.add_channel(name, cloud.channel(name).dtype)
        out.resize(total)
    k = len(grouped)
    optrs = out.device_ptrs() if out is not None else {}
    srcs = (C.c_void_p * k)(*[t.data_ptr() if t.numel() else None for t, _ in grouped.values()])
    dsts = (C.c_void_p * k)(*[optrs.get(name) if total else None for name in grouped])
    elems = (C.c_int32 * k)(*[_ELEM_BYTES[kind] for _, kind in grouped.values()])
    # (a rank that could not allocate announces no room: every rank refuses the round together)
    A.check(L.pcr_hip_comm_alltoallv(comm, k, srcs, dsts, elems, send, total if out is not None else 0, None, stream))
    if out is None:
        raise MemoryError("route_cloud: cannot allocate the routed cloud on the device")
    return out


def _route_cloud_on_current_stream(cloud, grid_cfg, blocks, rank, world, group, stream, comm=None):
    import pcr
    counts, grouped = partition_cloud(cloud, grid_cfg, blocks, stream)
    if comm is not None:
        return _alltoallv_native(comm, cloud, counts, grouped, world, stream)
    stage = dist.get_backend(group) == "gloo"
    dev = torch.device("cuda", torch.cuda.current_device())
    send = torch.tensor(counts, dtype=torch.int64, device="cpu" if stage else dev)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    recv_counts = [int(c) for c in recv.cpu().tolist()]
    total = sum(recv_counts)
    out = pcr.PointCloud.create(max(total, 1), pcr.MemoryLocation.Device)
    if out is None:
        raise MemoryError("route_cloud: cannot allocate the routed cloud on the device")
    for name in cloud.channel_names():
        out.add_channel(name, cloud.channel(name).dtype)
    out.resize(total)
    optrs = out.device_ptrs()
    for name, (src, kind) in grouped.items():
        view = pcr.DeviceArrayView(optrs[name], (max(total, 1),), _TYPESTR[kind], owner=out)
        dst = torch.as_tensor(view, device="cuda")[:total]
        if stage:
            buf = torch.empty(total, dtype=src.dtype)
            dist.all_to_all_single(buf, src.cpu(), recv_counts, counts, group=group)
            dst.copy_(buf)
        else:
            dist.all_to_all_single(dst, src, recv_counts, counts, group=group)
    return out


def _native_comm(rank, world, device, group=None):
    """pcr_hip_comm over RCCL: rank 0 makes the 128-byte id, torch.distributed only carries it."""
    import ctypes as C
    from . import _cabi as A
    L = A.lib()
    if not L.pcr_hip_comm_available():
        raise RuntimeError("ShardedPipeline(comm='native'): RCCL is not available to libpcr_hip.so")
    box = [None]
    if rank == 0:
        try:
            buf = (C.c_uint8 * 128)()
            A.check(L.pcr_hip_comm_unique_id(buf))
            box[0] = bytes(buf)
        except Exception as exc:                 # the others are waiting in the broadcast: they get the reason instead of the id
            box[0] = ("error", str(exc))
    dist.broadcast_object_list(box, src=0, group=group)
    if isinstance(box[0], tuple):
        raise RuntimeError("ShardedPipeline(comm='native'): rank 0 could not make the communicator's id: " + box[0][1])
    ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
    handle = C.c_void_p()
    A.check(L.pcr_hip_comm_create(C.byref(handle), ident, rank, world, int(device)))
    return handle


class ShardedPipeline:
    """pcr.Pipeline on this rank's row block + the halo exchange.  Usage (one process per GPU):

        sp = ShardedPipeline(cfg, rank, world)        # cfg.grid describes the WHOLE grid
        sp.ingest(cloud); ...; sp.finalize()           # finalize() = exchange + local finalize
        sp.result()                                    # rows [r0, r1) of every band
    """

    def __init__(self, cfg, rank, world, device_id=None, align=1, group=None, comm="torch"):
        """comm = "torch": the exchange runs through torch.distributed (nccl = RCCL on device tensors; gloo stages through
        host memory -- CPU tests and one-GPU rehearsals).  comm = "native": through the library's own pcr_hip_comm_*
        (include/pcr_hip.h: ncclSend / ncclRecv to rank +- 1 on the engine's stream, merged by a HIP kernel), bootstrapped
        by a unique id that rank 0 makes and torch.distributed merely carries to the other ranks."""
        import pcr
        if comm not in ("torch", "native"):
            raise ValueError("ShardedPipeline: comm must be 'torch' or 'native'")
        self.rank, self.world, self.group = rank, world, group
        self.comm_kind = comm
        self._comm = None
        self.grid = cfg.grid
        self.exchange_ms = None              # set by exchange(timed=True)
        self.blocks = [row_block(r, world, cfg.grid.height, align) for r in range(world)]
        self.own = self.blocks[rank]
        cfg.shard_row_begin, cfg.shard_row_end = self.own
        # ONE file for the whole grid, written by rank 0 from the gathered strips (the reference writes one file,
        # src/engine/pipeline.cpp:1351-1361) -- not a strip per rank under the same name
        self.output_path, cfg.output_path = cfg.output_path, ""
        self._state_dir = cfg.state_dir
        self._rtypes = [r.type for r in cfg.reductions]
        if device_id is not None:
            cfg.cuda_device_id = device_id
        self.pipe = pcr.Pipeline.create(cfg)
        if self.pipe is None:
            raise RuntimeError("Pipeline.create failed: " + pcr.pipeline_create_error())
        self.width = cfg.grid.width
        self.halo = self.pipe.halo_rows()
        self._line_hl_groups = any(r.glyph.type == pcr.GlyphType.Line and r.glyph.half_length_channel
                                   for r in cfg.reductions)
        self._views = None
        self._touched_ro = None
        self._keep_union = None
        # Touched flags are per reference tile.  When every block edge falls on a tile-row boundary no tile is
        # shared between ranks and the flags are purely local: a Point-glyph run then needs no collective at all.
        th = cfg.grid.tile_height
        self.tiles_local = all(b0 % th == 0 or b0 >= cfg.grid.height for b0, _ in self.blocks)
        # Feasibility is decided from EVERY rank's block (the same cfg gives the same blocks on every rank), so that the
        # constructor raises on every rank or on none: a rank that owns no rows, or a block shorter than the halo its
        # neighbours keep, cannot take part in the neighbour exchange.  (The torch exchange re-checks per call, the native
        # one agrees on the geometry inside every pcr_hip_comm_halo_reduce; this is the early, cheap refusal.)
        if world > 1 and not self.tiles_local and self.halo > 0:
            for r, (b0, b1) in enumerate(self.blocks):
                if b1 - b0 <= 0 or b1 - b0 < self.halo:
                    raise ValueError(f"ShardedPipeline: rank {r} would own {b1 - b0} rows with a halo of {self.halo} rows: "
                                     "use fewer ranks, a smaller radius or tile-aligned blocks (refused on every rank)")
        if comm == "native" and world > 1:
            self._comm = _native_comm(rank, world, cfg.cuda_device_id, group)

    def close(self):
        if self._comm is not None:
            from . import _cabi as A
            A.lib().pcr_hip_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _plane_tensors(self):
        if self._views is None:
            import pcr
            rows = self.pipe.state_row_count()
            self._views = []
            # only planes a glyph footprint can spill from take part in the halo reduce (a Point plane's halo rows
            # never receive anything)
            for (ptr, kind, _group), reach in zip(self.pipe.state_planes(), self.pipe.plane_reach_rows()):
                if reach <= 0:
                    continue
                view = pcr.DeviceArrayView(ptr, (rows, self.width), "<f4", owner=self.pipe)
                self._views.append((torch.as_tensor(view, device="cuda"), kind))
            ptr, tx, ty = self.pipe.tile_touched_ptr()
            self._touched = torch.as_tensor(pcr.DeviceArrayView(ptr, (ty * tx,), "<i4", owner=self.pipe),
                                            device="cuda")
        return self._views

    def ingest(self, cloud):
        """`cloud` holds (a superset of) the points this rank owns: the engine keeps the points whose centre row is
        in the owned block and ignores the rest.

        A Line group with a per-point half_length channel can need more halo rows than the shard keeps; every rank's
        need is reduced (MAX) BEFORE anything is accumulated, so that all ranks refuse the round together -- one rank
        raising alone would leave the others waiting in the next collective."""
        self._agree_line_reach(cloud)
        self.pipe.ingest(cloud)

    def _agree_line_reach(self, cloud):
        """MAX over the ranks of the rows this round's Line segments need beyond their centre row; raises on EVERY rank
        when that exceeds the halo.  Collective (when the pipeline has a Line group with a half_length channel and the
        blocks cut reference tiles): called by ingest() and, after routing, by ingest_unrouted()."""
        if not (self.world > 1 and self._line_hl_groups and not self.tiles_local):
            return
        try:
            need = int(self.pipe.line_reach_rows(cloud))
        except Exception:
            need = 2 ** 31 - 1                   # a rank whose query failed still takes part: no halo meets this need
        if self._comm is not None:
            import ctypes as C
            from . import _cabi as A
            word = C.c_int32(need)
            A.check(A.lib().pcr_hip_comm_agree_max_i32(self._comm, C.byref(word), self.pipe.stream_ptr()))
            need = int(word.value)
        else:
            cpu = dist.get_backend(self.group) == "gloo"
            t = torch.tensor([need], dtype=torch.int32, device="cpu" if cpu else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            need = int(t.item())
        if need > self.halo:
            raise RuntimeError(
                f"pipeline: a Line segment of this round reaches {need} rows beyond its centre row on some rank, but "
                f"the row-block shards keep a halo of {self.halo} rows; set PipelineConfig.shard_halo_rows >= {need} "
                "on every rank, or use tile-aligned row blocks (refused on every rank, nothing was accumulated)")

    def _engine_stream(self):
        ptr = self.pipe.stream_ptr()
        if ptr:
            return ptr, torch.cuda.stream(torch.cuda.ExternalStream(ptr))
        self.pipe.synchronize()
        return 0, torch.cuda.stream(torch.cuda.current_stream())

    def ingest_unrouted(self, cloud):
        """`cloud` is an ARBITRARY shard of the whole cloud (e.g. one file chunk per rank): its points are grouped
        by owner on the device, travel to their owners (all-to-all), and each rank ingests what it receives.
        Collective: every rank calls it once per round, with an empty cloud if it has nothing to contribute."""
        if self.world == 1:
            return self.pipe.ingest(cloud)
        import pcr
        if cloud.location() != pcr.MemoryLocation.Device:
            cloud = cloud.to_device()
        ptr, ctx = self._engine_stream()
        with ctx:
            mine = route_cloud(cloud, self.grid, self.blocks, self.rank, self.world, self.group, ptr, comm=self._comm)
            if not ptr:
                torch.cuda.current_stream().synchronize()
        self._agree_line_reach(mine)         # after routing: every rank asks about the points it will really ingest
        self.pipe.ingest(mine)
        self.pipe.synchronize()              # `mine` is freed on return
        return mine.count()

    def exchange(self, timed=False):
        """Halo reduce + touched-tile union.  No-op for a single rank."""
        # tiles_local: glyph footprints are clipped to the reference tile of their centre cell (Q4), so with
        # tile-aligned blocks nothing ever lands in a neighbour's rows -- no halo to reduce either
        if self.world == 1 or self.tiles_local:
            return
        if self.halo == 0 and self._comm is None:
            return self._exchange_flags_only(timed)
        planes = self._plane_tensors()
        if self._comm is not None:
            return self._exchange_native(planes, timed)
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
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if self.halo > 0 and planes:
                exchange_halos(planes, self.own, self.pipe.state_row_begin(), self.halo,
                               self.rank, self.world, blocks=self.blocks, group=self.group)
            if not self.tiles_local:
                allreduce_touched(self._touched, self.group)
            if timed:
                e1.record()
        if not ptr:
            torch.cuda.current_stream().synchronize()
        if timed:
            e1.synchronize()
            self.exchange_ms = e0.elapsed_time(e1)

    def _exchange_flags_only(self, timed):
        """A pipeline without glyph planes (halo 0) exchanges nothing but the touched-tile flags.  They are all-reduced in a COPY
        and merged back by Pipeline.merge_touched: the pipeline's planes and flags are never handed out for writing, so the
        bands its scatter may have stored stay valid unless another rank really touched a tile this one did not (decided on
        the device: no host synchronisation)."""
        import pcr
        if self._touched_ro is None:
            ptr, tx, ty = self.pipe.tile_touched_ptr(readonly=True)
            self._touched_ro = torch.as_tensor(pcr.DeviceArrayView(ptr, (ty * tx,), "<i4", owner=self.pipe), device="cuda")
        ptr = self.pipe.stream_ptr()
        if ptr:
            ctx = torch.cuda.stream(torch.cuda.ExternalStream(ptr))
        else:
            self.pipe.synchronize()
            ctx = torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            union = self._touched_ro.clone()
            allreduce_touched(union, self.group)
            self.pipe.merge_touched(union.data_ptr())           # (on the pipeline's stream = this one, or after the sync below)
            if timed:
                e1.record()
        if not ptr:
            torch.cuda.current_stream().synchronize()
        if timed:
            e1.synchronize()
            self.exchange_ms = e0.elapsed_time(e1)
        self._keep_union = union                                # alive until the merge kernel has run (next exchange / finalize sync)

    def _exchange_native(self, planes, timed):
        """pcr_hip_comm_halo_reduce + pcr_hip_comm_allreduce_max_u32 on the engine's stream (no torch in the data path)."""
        import ctypes as C
        from . import _cabi as A
        L = A.lib()
        stream = self.pipe.stream_ptr()
        if timed:
            ev = [C.c_void_p(), C.c_void_p()]
            for e in ev:
                A.check(L.pcr_hip_event_create(C.byref(e)))
            A.check(L.pcr_hip_event_record(ev[0], stream))
        if self.halo > 0 and planes:
            arr = (A.HaloPlane * len(planes))()
            for k, (t, kind) in enumerate(planes):
                arr[k].d_plane, arr[k].kind = t.data_ptr(), kind
            A.check(L.pcr_hip_comm_halo_reduce(self._comm, arr, len(planes), self.width, self.pipe.state_row_begin(),
                                               self.pipe.state_row_count(), self.own[0], self.own[1], self.halo, stream))
        A.check(L.pcr_hip_comm_allreduce_max_u32(self._comm, self._touched.data_ptr(), self._touched.numel(), stream))
        if timed:
            A.check(L.pcr_hip_event_record(ev[1], stream))
            A.check(L.pcr_hip_stream_synchronize(stream))
            ms = C.c_float()
            A.check(L.pcr_hip_event_elapsed_ms(ev[0], ev[1], C.byref(ms)))
            self.exchange_ms = float(ms.value)
            for e in ev:
                L.pcr_hip_event_destroy(e)

    def collectives_per_step(self):
        """What exchange() issues on this rank: (point-to-point halo messages, all-reduces)."""
        if self.world == 1 or self.tiles_local:
            return 0, 0
        nplanes = sum(1 for r in self.pipe.plane_reach_rows() if r > 0)
        neighbours = (1 if self.rank > 0 else 0) + (1 if self.rank < self.world - 1 else 0)
        return (2 * neighbours * nplanes if self.halo > 0 else 0), 1

    def halo_bytes_per_step(self):
        """Bytes this rank SENDS in one exchange: halo rows x width x 4 per plane and neighbour."""
        p2p, _ = self.collectives_per_step()
        return (p2p // 2) * self.halo * self.width * 4

    def finalize(self, timed=False, wait=True):
        """exchange + local finalize.  wait=False: Pipeline.finalize_async (a device-resident result is complete after
        pipe.synchronize() / a device synchronisation)."""
        self.exchange(timed)
        if wait or self.output_path:
            self.pipe.finalize()
        else:
            self.pipe.finalize_async()
        if self.output_path:
            import pcr
            whole = self.gather(0)
            if self.rank == 0:
                pcr.write_geotiff(self.output_path, whole, self.grid)

    def save_state(self, directory=""):
        """`.pcrt` checkpoint of the sharded pipeline (collective): the exchange first (what a rank's apron rows hold belongs
        in its neighbour's tiles).  Blocks of whole reference-tile rows (align = tile_height): every rank writes the tiles it
        owns, their union is the checkpoint.  Blocks that cut tiles: a tile has two owners and a file holds a whole tile, so
        the planes' owned rows are gathered to rank 0, which writes the whole grid's tiles.  Either way the files are those an
        unsharded pipeline would write (and reads back: resume at any world size)."""
        self.exchange()
        if self.world == 1 or self.tiles_local:
            return self.pipe.save_state(directory)
        import os
        import numpy as np
        import pcr
        directory = directory or self._state_dir
        if not directory:
            raise RuntimeError("pipeline: no state directory given")
        rows_state, s0 = self.pipe.state_row_count(), self.pipe.state_row_begin()
        o0, o1 = self.own
        keys, strips = [], []
        for ptr, kind, group in self.pipe.state_planes():
            t = torch.as_tensor(pcr.DeviceArrayView(ptr, (rows_state, self.width), "<f4", owner=self.pipe), device="cuda")
            keys.append((group, kind))
            strips.append(t[o0 - s0:o1 - s0])
        planes = self._gather_rows(0, strips)
        ptr, tx, ty = self.pipe.tile_touched_ptr(readonly=True)                  # the union, after the exchange
        touched = torch.as_tensor(pcr.DeviceArrayView(ptr, (ty, tx), "<i4", owner=self.pipe), device="cuda").cpu().numpy()
        if self.rank != 0:
            return
        T = pcr.ReductionType
        fields = {T.Sum: (PLANE_SUM,), T.Count: (PLANE_WGT,), T.Max: (PLANE_MAX,), T.Min: (PLANE_MIN,)}       # builtin_ops.h state layouts
        g = self.grid
        groups = self.pipe.reduction_groups()
        for r, (grp, rtype) in enumerate(zip(groups, self._rtypes)):
            state = np.stack([planes[keys.index((grp, k))] for k in fields.get(rtype, (PLANE_SUM, PLANE_WGT))])
            rdir = directory if len(groups) == 1 else os.path.join(directory, f"reduction_{r}")
            os.makedirs(rdir, exist_ok=True)
            for row in range(ty):
                for col in range(tx):
                    if not touched[row, col]:
                        continue
                    r0, c0 = row * g.tile_height, col * g.tile_width
                    tile = state[:, r0:min(r0 + g.tile_height, g.height), c0:min(c0 + g.tile_width, g.width)]
                    pcr.write_tile_state(pcr.tile_state_filename(rdir, row, col), row, col, np.ascontiguousarray(tile), rtype)

    def load_state(self, directory=""):
        """Every rank takes the tiles of its own rows from `directory` (also: PipelineConfig.resume at create)."""
        self.pipe.load_state(directory)

    def result(self):
        """This rank's STRIP: rows [own[0], own[1]) of every band.  gather() assembles the whole grid on one rank."""
        return self.pipe.result()

    def _gather_rows(self, dst_rank, strips):
        """Collective: `strips` = this rank's owned rows of some arrays (tensors [rows, W]: device tensors, or host tensors
        over gloo); returns, on dst_rank, the list of assembled [H, W] float32 numpy arrays (None elsewhere).  The strips in
        rank order are the grid's rows in order.  comm = "native": pcr_hip_comm_gatherv, device to device, up to eight arrays
        per grouped round; comm = "torch": send / recv (nccl: from HBM; gloo: staged through host memory)."""
        import numpy as np
        W, H = self.width, self.grid.height
        root = self.rank == dst_rank
        rows = self.own[1] - self.own[0]
        out = []
        if self._comm is not None:
            import ctypes as C
            from . import _cabi as A
            L = A.lib()
            stream, ctx = self._engine_stream()
            per_round = max(1, min(8, (4 << 30) // max(H * W * 4, 1)))
            with ctx:
                for b0 in range(0, len(strips), per_round):
                    part = [t if t.is_cuda else t.cuda() for t in strips[b0:b0 + per_round]]
                    k = len(part)
                    land = torch.empty((k, H, W), dtype=torch.float32, device="cuda") if root else None
                    srcs = (C.c_void_p * k)(*[t.data_ptr() for t in part])
                    dsts = (C.c_void_p * k)(*[land[a].data_ptr() if root else None for a in range(k)])
                    elems = (C.c_int32 * k)(*([4] * k))
                    A.check(L.pcr_hip_comm_gatherv(self._comm, k, srcs, dsts, elems, rows * W, H * W if root else 0, None,
                                                   dst_rank, stream))
                    if root:
                        host = land.cpu()                    # (on this stream: after the receives)
                        out += [host[a].numpy() for a in range(k)]
                if not stream:
                    torch.cuda.current_stream().synchronize()
            return out if root else None
        stage = dist.get_backend(self.group) == "gloo"
        if stage and not any(t.is_cuda for t in strips):
            import contextlib
            ctx = contextlib.nullcontext()           # host arrays over gloo: no device in the path at all
        else:
            _, ctx = self._engine_stream()
        with ctx:
            for t in strips:
                mine = t.cpu() if stage else (t if t.is_cuda else t.cuda())
                if root:
                    full = torch.empty((H, W), dtype=torch.float32, device="cpu" if stage else "cuda")
                    ops = [dist.P2POp(dist.irecv, full[b0:b1], r, self.group)
                           for r, (b0, b1) in enumerate(self.blocks) if r != self.rank and b1 > b0]
                    for req in (dist.batch_isend_irecv(ops) if ops else []):
                        req.wait()
                    full[self.own[0]:self.own[1]] = mine
                    out.append(full.cpu().numpy())
                elif rows > 0:
                    for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, mine.contiguous(), dst_rank, self.group)]):
                        req.wait()
        return out if root else None

    def gather(self, dst_rank=0):
        """Collective, after finalize(): every rank's strip travels to `dst_rank`, which returns ONE host pcr.Grid of the whole
        grid (the strips in rank order are the grid's rows in order); the other ranks return None.  The reference's result()
        is one grid (src/engine/pipeline.cpp:1175-1186).  Transport: _gather_rows."""
        import numpy as np
        import pcr
        res = self.pipe.result()
        if res is None:
            raise RuntimeError("ShardedPipeline.gather: finalize() first")
        if not 0 <= dst_rank < self.world:
            raise ValueError("ShardedPipeline.gather: destination rank outside [0, world)")
        nb, W, H = res.num_bands(), self.width, self.grid.height
        root = self.rank == dst_rank
        rows = self.own[1] - self.own[0]
        on_device = res.location() == pcr.MemoryLocation.Device

        def strip(b):                              # this rank's band b: a zero-copy device view, or the host band
            ptr = self.pipe.result_band_device_ptr(b) if hasattr(self.pipe, "result_band_device_ptr") else 0
            if ptr and (on_device or self._comm is not None or dist.get_backend(self.group) != "gloo"):
                return torch.as_tensor(pcr.DeviceArrayView(ptr, (rows, W), "<f4", owner=self.pipe), device="cuda")
            return torch.from_numpy(np.array(res.band_array(b)))

        whole = pcr.Grid.create(W, H, [res.band_desc(b) for b in range(nb)]) if root else None
        if self.world == 1:
            for b in range(nb):
                whole.set_band_array(b, strip(b).cpu().numpy())
            return whole
        bands = self._gather_rows(dst_rank, [strip(b) for b in range(nb)])
        if root:
            for b in range(nb):
                whole.set_band_array(b, bands[b])
        return whole
