"""Axis-0 slab decomposition of one large volume across the GPUs of a node (SURVEY.md section 8e;
new capability, the reference is single-GPU only).

Rank r owns sample planes [x0, x1) of the [rx, ry, rz] grid and every cell whose lower corner lies in
them.  The cells of its last layer need plane x1, which lives on rank r+1: ONE halo plane is
received over RCCL (torch.distributed "nccl" backend = RCCL over xGMI, point-to-point send/recv,
4 MiB at 1024^2 fp32).  Vertices on edges inside that halo plane are owned by rank r+1, so after the
local single-pass extraction rank r+1 also ships the vertex-id records of its plane 0 (ry*ncz*8 B,
128 KiB at 1024^2) and every rank learns the global vertex base of its neighbour from an
all-gather of the per-rank counts.  No other collective touches the data path.

    phase A   halo plane  r+1 -> r        (send/recv)
    phase B   local one-pass extraction   (vertices final, faces counted)
    phase C   counts all-gather; plane-0 records r+1 -> r
    phase D   local face emission with global vertex ids

The compute backend is pluggable so the orchestration can be tested without a GPU: `HipBackend`
drives the C ABI (include/p3d_mc.h); tests inject a CPU stand-in built on the oracle.
"""
from typing import List, Optional, Sequence, Tuple

import torch


def slab_bounds(rx: int, world: int) -> List[Tuple[int, int]]:
    """[x0, x1) per rank, as even as possible; every rank gets at least one plane."""
    assert world >= 1 and rx >= world, "need at least one plane per rank"
    base, rem = divmod(rx, world)
    out, x = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((x, x + n))
        x += n
    return out


# ---------------------------------------------------------------------------------------------
# backends
# ---------------------------------------------------------------------------------------------
class HipBackend:
    """Local extraction through the C ABI on the rank's own GPU."""

    def __init__(self, device):
        from . import capi
        self.capi = capi
        self.device = device
        self._cap = None  # output-size hint from the previous call
        self._ws = None

    def _prepare(self, grid, x_origin, halo, thresh=None, lower=None, upper=None, full_res=None):
        c = self.capi
        rx, ry, rz = grid.shape
        if getattr(self, "_ws_shape", None) != (rx, ry, rz):
            nbytes = c.workspace_bytes(rx, ry, rz)
            if self._ws is None or self._ws.numel() < nbytes:
                self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws_shape = (rx, ry, rz)
            ptr, self._plane_bytes = c.plane_records(self._ws, rx, ry, rz, 0)
            self._rec0_off = ptr - self._ws.data_ptr()
        # everything the parts of this extraction have in common, marshalled once (capi.FusedCaller)
        self._call = c.FusedCaller(grid, thresh, lower, upper, self._ws, full_res) if thresh is not None else None
        capv = self._cap if self._cap is not None else max(4096, rx * ry * rz // 16)
        self._verts = torch.empty((capv, 3), dtype=torch.float32, device=self.device)
        self._scratch = torch.empty((c.scratch_rows_for(capv), 3), dtype=torch.float32, device=self.device)
        # rc: the all-gathered vertex counts on the device, [world] int64 or [world, k] with the counts in column 0
        self._mk = lambda part=0, split=0, vb=0, hb=0, rc=None, rank=0: c.Slab(
            1 if halo else 0, part, vb, hb, x_origin, split, rc.data_ptr() if rc is not None else None, rank,
            rc.stride(0) if rc is not None else 0)

    def begin_interior(self, grid, thresh, lower, upper, full_res, x_origin, halo, split):
        """Stream planes [0, split): they do not touch the halo plane, so this can run while it is in flight."""
        self._prepare(grid, x_origin, halo, thresh, lower, upper, full_res)
        self._split = split
        self._call.fused(self._mk(1, split), self._verts, self._scratch, None)

    def stream_rest(self, grid, thresh, lower, upper, full_res, x_origin, halo):
        """Stream the planes not yet streamed and leave V and the id prefixes in the workspace header (no finalize):
        what the collectives need is on the device now, so they can be enqueued and travel during `finalize`."""
        c = self.capi
        split = getattr(self, "_split", 0)
        if not split:
            self._prepare(grid, x_origin, halo, thresh, lower, upper, full_res)
        slab = self._mk(3, split)
        self._exported = None
        if getattr(self, "export_first_plane", False):   # (rank > 0: the previous rank's halo records, in the header's launch)
            self._exported = torch.empty(self._plane_bytes, dtype=torch.uint8, device=self.device)
            slab.export_first_plane_to = self._exported.data_ptr()
        self._call.fused(slab, self._verts, self._scratch, None)
        self._state = (grid, thresh, lower, upper, full_res, self._ws, None)

    def header_vertex_count(self):
        """This rank's vertex count as a device tensor (the first int64 of the workspace)."""
        return self._ws[:8].view(torch.int64)

    def header_words(self):
        """The first three int64 of the workspace header -- (V, unused here, flags) -- as a device tensor: the input of
        the all-gather, so that every rank learns every rank's vertex count AND overflow flags."""
        return self._ws[:24].view(torch.int64)

    def launch_finalize(self, defer_totals=False):
        """Part 4: face count + the first slices of the vertex compaction; V and F go to the host mailbox -- unless
        `defer_totals`: then the part 5 that finish_on_device enqueues right behind reports them from the first block of
        its face launch (p3d_mc_slab.defer_totals: no totals launch on the stream)."""
        slab = self._mk(4, getattr(self, "_split", 0))
        self._deferred = bool(defer_totals) and getattr(self, "_capf", None) is not None
        slab.defer_totals = 1 if self._deferred else 0
        self._call.fused(slab, self._verts, self._scratch, None)
        self._split = 0
        self._slab = self._mk()

    def _read_totals(self):
        """Wait for V and F; after a vertex-capacity overflow rewrite the vertices exactly sized (ids stay valid)."""
        c = self.capi
        grid, thresh, lower, upper, full_res, ws, _ = self._state
        verts = self._verts
        nv, nf, over = self._call.read_counts()
        # bit 1: a vertex region numbered more than 2^26 vertices, its ids are ambiguous.  Not raised HERE: the other
        # ranks would go on into the collectives and hang -- the orchestration spreads the flag to all ranks first
        # (SlabExtractor.extract) and every rank raises.
        self.id_overflow = bool(over & 2)
        overflow = nv > verts.shape[0] or (over & 1)
        if overflow:
            verts = torch.empty((nv, 3), dtype=torch.float32, device=self.device)
            self._call.emit(self._slab, verts, None)
        self._cap = nv + nv // 8 + 4096
        self._state = (grid, thresh, lower, upper, full_res, ws, nf)
        return nv, nf, verts[:nv], overflow

    def finalize(self):
        """Host path: totals first (the caller all-gathers them on the host), faces later.  The vertices are complete
        (in stream order) once the faces have been written: the rest of the copy rides in that launch."""
        self.launch_finalize()
        nv, nf, verts, overflow = self._read_totals()
        self._copy_pending = not overflow
        if overflow:
            self._scratch = None
        return nv, nf, verts

    def finish_on_device(self, rank_counts, rank):
        """Device path: the face launch is enqueued with a capacity guess from the previous call BEFORE the host has
        seen the totals, so no host round trip sits between the counting kernel and the faces; the totals are read
        afterwards and the buffers narrowed (exact re-emission if the guess was too small)."""
        c = self.capi
        grid, thresh, lower, upper, full_res, ws, _ = self._state
        capf = getattr(self, "_capf", None)
        if capf is None:  # first call: no guess yet -- totals first (part 4 is already enqueued: launch_finalize), then faces
            nv, nf, verts, overflow = self._read_totals()
            self._copy_pending = not overflow
            if overflow:
                self._scratch = None
            faces = self._emit_faces((0, 0, rank_counts, rank))
        else:
            faces = torch.empty((capf, 3), dtype=torch.int32, device=self.device)
            slab5 = self._mk(5, 0, 0, 0, rank_counts, rank)
            slab5.defer_totals = 1 if getattr(self, "_deferred", False) else 0
            self._call.fused(slab5, self._verts, self._scratch, faces)
            nv, nf, verts, overflow = self._read_totals()
            self._scratch = None
            self._copy_pending = False
            if self.id_overflow:
                pass   # (the caller raises; the ids are ambiguous, nothing to re-emit)
            elif nf > capf:
                faces = torch.empty((nf, 3), dtype=torch.int32, device=self.device)
                self._call.emit(self._mk(0, 0, 0, 0, rank_counts, rank), None, faces)
            else:
                faces = faces[:nf] if 2 * nf >= capf else faces[:nf].clone()
        self._capf = nf + nf // 8 + 4096
        return nv, nf, verts, faces

    def count_and_vertices(self, grid, thresh, lower, upper, full_res, x_origin, halo):
        self.stream_rest(grid, thresh, lower, upper, full_res, x_origin, halo)
        return self.finalize()

    def _plane_view(self, plane):
        ws = self._state[5]
        off = self._rec0_off + plane * self._plane_bytes   # (records of plane p: p3d_mc_plane_records; planes are back to back)
        return ws[off:off + self._plane_bytes]

    def export_first_plane_records(self):
        if getattr(self, "_exported", None) is not None:   # (written by part 3's header launch)
            return self._exported
        out = torch.empty(self._plane_bytes, dtype=torch.uint8, device=self.device)
        return self._call.export_plane_records(0, out)

    def halo_records_buffer(self):
        return self._plane_view(self._state[0].shape[0] - 1)

    def _emit_faces(self, slab_args):
        """slab_args -> (vb, hb, rank_counts, rank).  One launch writes the faces and finishes the vertex copy
        (part 5); after a capacity overflow the vertices were already rewritten and only the faces remain."""
        grid, thresh, lower, upper, full_res, ws, nf = self._state
        faces = torch.empty((nf, 3), dtype=torch.int32, device=self.device)
        if self._copy_pending:
            self._call.fused(self._mk(5, 0, *slab_args), self._verts, self._scratch, faces)
            self._copy_pending = False
            self._scratch = None
        else:
            self._call.emit(self._mk(0, 0, *slab_args), None, faces)
        return faces

    def faces(self, vertex_id_base, halo_vertex_id_base):
        return self._emit_faces((vertex_id_base, halo_vertex_id_base, None, 0))

    def faces_from_rank_counts(self, rank_counts, rank):
        """Same, with the id bases derived ON THE DEVICE from the all-gathered [world] int64 vertex counts: the host
        does not wait for the other ranks (include/p3d_mc.h: p3d_mc_slab.rank_counts)."""
        return self._emit_faces((0, 0, rank_counts, rank))


# ---------------------------------------------------------------------------------------------
# orchestration
# ---------------------------------------------------------------------------------------------
class SlabResult:
    """vertices: [V_r, 3] f32, already in bounding-box coordinates of the FULL grid
    faces:    [F_r, 3] i32, GLOBAL vertex ids (vertices of all ranks concatenated in rank order)
    vertex_base: this rank's first global vertex id; when the extraction kept the all-gathered vertex counts on the
    device it is computed from them on first access (a device-to-host copy, i.e. a synchronisation).
    counts: (V, F) of every rank -- host path only."""

    def __init__(self, vertices, faces, vertex_base=None, counts=None, rank=None, rank_counts=None, check=None):
        self.vertices, self.faces = vertices, faces
        self._base, self.counts, self._rank, self._rank_counts = vertex_base, counts, rank, rank_counts
        self._check = check

    def check_total(self):
        """Device path: wait for this extraction's gathered vertex counts and raise OverflowError if their sum does not
        fit int32 face indices (the extractor does the same, without waiting, when its next extraction starts)."""
        if self._check is not None:
            self._check()

    @property
    def vertex_base(self) -> int:
        if self._base is None:
            rc = self._rank_counts if self._rank_counts.dim() == 1 else self._rank_counts[:, 0]
            self._base = int(rc[:self._rank].sum()) if self._rank else 0
        return self._base

    def __iter__(self):  # (vertices, faces) unpacking like the single-GPU call
        return iter((self.vertices, self.faces))


class SlabExtractor:
    def __init__(self, shape: Sequence[int], rank: int, world: int, device, dtype=torch.float32, backend=None,
                 hold_planes: int = 2):
        self.shape = tuple(int(s) for s in shape)
        # the device path derives a slab's id bases from the gathered counts with one lane of a wave per rank
        # (include/p3d_mc.h: p3d_mc_slab.rank_counts serves ranks 0..63).  Refused HERE, on every rank alike and before any
        # collective: the C ABI would refuse ranks >= 64 only inside finish_on_device, after the all-gather and the record
        # exchange were enqueued, and the ranks below 64 would wait for them in the next collective for ever.
        if world > 64 and (backend is None or hasattr(backend, "faces_from_rank_counts")):
            raise ValueError(f"SlabExtractor serves at most 64 ranks on the device path (world = {world}): "
                             "use fewer, thicker slabs")
        self.hold_planes = max(1, int(hold_planes))   # local planes kept back for the launch that needs the halo plane
        self.rank, self.world = rank, world
        self.device = device
        self.x0, self.x1 = slab_bounds(self.shape[0], world)[rank]
        self.n = self.x1 - self.x0
        self.has_halo = rank < world - 1
        rx, ry, rz = self.shape
        self.grid = torch.empty((self.n + (1 if self.has_halo else 0), ry, rz), dtype=dtype, device=device)
        self.backend = backend if backend is not None else HipBackend(device)
        if rank > 0 and hasattr(self.backend, "stream_rest"):
            self.backend.export_first_plane = True   # part 3 writes plane 0's dense records with the header (one launch)
        # phase tracing (bench.py --stages): events on the current stream at the phase boundaries of extract(); the
        # span between two marks is the GPU time of what was enqueued between them, waits on collectives included
        self.trace = False
        self._marks = []

    def _mark(self, name):
        if self.trace and self.grid.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._marks.append((name, ev))

    def phase_times_ms(self):
        """{phase: ms} of the last traced extract() (synchronises)."""
        if len(self._marks) < 2:
            return {}
        self._marks[-1][1].synchronize()
        return {b[0]: a[1].elapsed_time(b[1]) for a, b in zip(self._marks, self._marks[1:])}

    # -- data ---------------------------------------------------------------------------------
    def fill_local(self, gen):
        """gen(x0, x1) -> [x1-x0, ry, rz] tensor: the rank synthesises / loads only its own planes."""
        self.grid[:self.n].copy_(gen(self.x0, self.x1))

    # -- phases (the in-process test harness calls them in lock step) ---------------------------
    def halo_send_buffer(self) -> Optional[torch.Tensor]:
        return self.grid[0] if self.rank > 0 else None

    def halo_recv_buffer(self) -> Optional[torch.Tensor]:
        return self.grid[self.n] if self.has_halo else None

    def interior_split(self) -> int:
        """First plane of the part that needs the halo plane (0 = do not split): only the last cell layer reads it; the
        last `hold_planes` (default 2) local planes are held back for a short second streaming launch."""
        return self.n - self.hold_planes if (self.has_halo and self.n >= 24 and hasattr(self.backend, "begin_interior")) else 0

    def phase_interior(self, thresh, lower, upper):
        """Optional: start streaming the planes that do not depend on the halo plane."""
        split = self.interior_split()
        if split:
            self.backend.begin_interior(self.grid, float(thresh), list(lower), list(upper), self.shape, self.x0,
                                        self.has_halo, split)
        return split

    def phase_extract(self, thresh, lower, upper):
        nv, nf, verts = self.backend.count_and_vertices(self.grid, float(thresh), list(lower), list(upper),
                                                        self.shape, self.x0, self.has_halo)
        self._nv, self._nf, self._verts = nv, nf, verts
        return nv, nf

    ID_OVERFLOW = ("a vertex region of the slab of rank %s numbered more than 2^26 vertices (include/p3d_mc.h, "
                   "p3d_mc_read_counts bit 1): use more ranks / thinner slabs")

    def id_overflow(self) -> bool:
        """This rank's last extraction handed out ambiguous vertex ids (every rank must learn it: extract())."""
        return bool(getattr(self.backend, "id_overflow", False))

    def records_send_buffer(self):
        return self.backend.export_first_plane_records() if self.rank > 0 else None

    def records_recv_buffer(self):
        return self.backend.halo_records_buffer() if self.has_halo else None

    def phase_faces(self, counts: List[Tuple[int, int]]) -> SlabResult:
        base = sum(c[0] for c in counts[:self.rank])
        total = sum(c[0] for c in counts)
        if total > 2 ** 31 - 1:
            raise OverflowError("global vertex count exceeds int32 face indices")
        halo_base = base + counts[self.rank][0]
        faces = self.backend.faces(base, halo_base)
        return SlabResult(self._verts, faces, base, counts)

    def _make_total_check(self, host, ev):
        """The deferred check of ONE extraction (its own pinned copy of the gathered header words and its own event), as
        a closure: a SlabResult keeps its own, whatever the extractor does afterwards."""
        state = {"done": False, "error": None}

        def check(wait: bool = True):
            if not state["done"]:
                if wait:
                    ev.synchronize()
                elif not ev.query():
                    return
                state["done"] = True
                rows = host.tolist()   # (one conversion: indexing a tensor element by element is microseconds each)
                bad = [r for r, row in enumerate(rows) if row[2] & 2]
                if bad:
                    state["error"] = OverflowError(self.ID_OVERFLOW % bad)
                elif sum(row[0] for row in rows) > 2 ** 31 - 1:
                    state["error"] = OverflowError("global vertex count exceeds int32 face indices")
            if state["error"] is not None:
                raise state["error"]
        return check

    def _check_pending_total(self, wait: bool = True):
        check = getattr(self, "_pending_check", None)
        if check is not None:
            self._pending_check = None
            check(wait)

    # -- the distributed call -----------------------------------------------------------------
    def extract(self, thresh, lower=None, upper=None) -> SlabResult:
        import torch.distributed as dist
        rx, ry, rz = self.shape
        lower = [0.0, 0.0, 0.0] if lower is None else lower
        upper = [rx, ry, rz] if upper is None else upper

        def shift_to_prev(send, recv):  # rank r sends to r-1, receives from r+1: nearest-neighbour only
            ops = []
            if send is not None:
                ops.append(dist.P2POp(dist.isend, send, self.rank - 1))
            if recv is not None:
                ops.append(dist.P2POp(dist.irecv, recv, self.rank + 1))
            return dist.batch_isend_irecv(ops) if ops else []

        # RCCL collectives are ordered after the work already enqueued on the current stream.  gloo (CPU transport, used
        # by the tests and the single-GPU dry run) reads device tensors without that ordering: synchronise for it.
        gloo_on_gpu = self.grid.is_cuda and dist.get_backend() == "gloo"

        def pre_comm():
            if gloo_on_gpu:
                torch.cuda.synchronize()

        # the halo plane travels while the interior planes are already being streamed
        self._marks = []
        self._mark("start")
        pre_comm()
        works = shift_to_prev(self.halo_send_buffer(), self.halo_recv_buffer())
        self.phase_interior(thresh, lower, upper)
        self._mark("interior planes streamed")
        for w in works:
            w.wait()
        self._mark("halo plane received (wait)")
        if hasattr(self.backend, "faces_from_rank_counts"):
            # device path: V and the id prefixes are in the workspace header as soon as the slab is streamed, so the
            # all-gather of V and the record exchange are enqueued BEFORE the face count / vertex compaction and
            # travel while those run; the other ranks' counts never visit the host (the face kernel derives its id
            # bases from the gathered tensor)
            be = self.backend
            be.stream_rest(self.grid, float(thresh), list(lower), list(upper), self.shape, self.x0, self.has_halo)
            # (V, -, flags) of every rank: the face kernel reads the counts with a stride of 3 (p3d_mc_slab), and every
            # rank learns every rank's overflow flags with them
            rank_counts = torch.empty((self.world, 3), dtype=torch.int64, device=self.grid.device)
            send_buf = self.records_send_buffer()  # (kept referenced until the transfer has completed)
            self._mark("last planes streamed + record export")
            pre_comm()
            # ASYNCHRONOUS: the plain form makes the current stream wait for the collective right here (torch's non-async
            # collectives end in work.wait()), i.e. the face count below would start only when the slowest rank's count had
            # arrived -- the all-gather's latency and the ranks' skew on every rank's critical path.  Waited for in front
            # of the face launch, the only consumer (round 6: it had been the plain form since round 3).
            gathered = dist.all_gather_into_tensor(rank_counts.view(-1), be.header_words(), async_op=True)
            self._mark("all-gather of vertex counts")
            rec_works = shift_to_prev(send_buf, self.records_recv_buffer())
            be.launch_finalize(defer_totals=True)   # (from the second call on: the faces follow with a capacity guess)
            self._mark("face count + early vertex copy")
            for w in rec_works:
                w.wait()
            if gathered is not None:
                gathered.wait()   # (a stream-level wait under RCCL: the host goes on)
            counts_ready = torch.cuda.Event() if self.grid.is_cuda else None
            if counts_ready is not None:
                counts_ready.record()
            self._mark("halo records received (wait)")
            del send_buf
            self._nv, self._nf, self._verts, faces = be.finish_on_device(rank_counts, self.rank)
            self._mark("faces + rest of vertex copy")
            # int32 guard on the GLOBAL vertex total (the host path checks it in phase_faces): the gathered counts live
            # on the device, so they are copied to pinned memory behind the work already enqueued and looked at when the
            # next extraction starts or when the caller asks (SlabResult.check_total) -- never a wait inside this call
            # (on a side stream, ordered only after the all-gather: inside the compute stream the 64-byte copy cost the
            #  stream ~10 us per extraction)
            self._check_pending_total()
            host = torch.empty((self.world, 3), dtype=torch.int64, pin_memory=True) if self.grid.is_cuda else None
            check = None
            if host is not None:
                if getattr(self, "_guard_stream", None) is None:
                    self._guard_stream = torch.cuda.Stream(device=self.grid.device)
                gs = self._guard_stream
                gs.wait_event(counts_ready)
                with torch.cuda.stream(gs):
                    host.copy_(rank_counts, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(gs)
                rank_counts.record_stream(gs)
                check = self._make_total_check(host, ev)
                self._pending_check = check
            if self.id_overflow():
                # this rank knows already; every collective of this extraction has been enqueued, and the other ranks
                # raise from the gathered flags before their next one (the deferred check at the start of extract())
                raise OverflowError(self.ID_OVERFLOW % self.rank)
            return SlabResult(self._verts, faces, rank=self.rank, rank_counts=rank_counts, check=check)
        nv, nf = self.phase_extract(thresh, lower, upper)
        mine = torch.tensor([nv, nf, 1 if self.id_overflow() else 0], dtype=torch.int64, device=self.grid.device)
        allc = [torch.empty_like(mine) for _ in range(self.world)]
        send_buf = self.records_send_buffer()
        pre_comm()
        dist.all_gather(allc, mine)
        gathered = torch.stack(allc).cpu()
        counts = [(int(c[0]), int(c[1])) for c in gathered]
        for w in shift_to_prev(send_buf, self.records_recv_buffer()):
            w.wait()
        bad = [r for r in range(self.world) if int(gathered[r, 2])]
        if bad:   # every rank raises, after the collectives of this extraction: nobody is left waiting
            raise OverflowError(self.ID_OVERFLOW % bad)
        return self.phase_faces(counts)


def extract_in_process(grid_full: torch.Tensor, world: int, thresh, lower=None, upper=None, device=None,
                       backend_factory=None) -> List[SlabResult]:
    """Run all `world` ranks of the slab algorithm sequentially in ONE process (copies stand in for the
    send/recv pairs).  Used to validate the multi-GPU path on a single device."""
    device = grid_full.device if device is None else device
    shape = tuple(grid_full.shape)
    lower = [0.0, 0.0, 0.0] if lower is None else lower
    upper = list(shape) if upper is None else upper
    exs = [SlabExtractor(shape, r, world, device, dtype=grid_full.dtype,
                         backend=backend_factory(r) if backend_factory else None) for r in range(world)]
    for e in exs:
        e.fill_local(lambda x0, x1: grid_full[x0:x1].to(device))
    for e in exs:  # interior planes first (in the distributed run this overlaps the halo transfer)
        e.phase_interior(thresh, lower, upper)
    for r in range(world - 1):  # phase A
        exs[r].halo_recv_buffer().copy_(exs[r + 1].halo_send_buffer())
    counts = [e.phase_extract(thresh, lower, upper) for e in exs]  # phase B
    if any(e.id_overflow() for e in exs):
        raise OverflowError(SlabExtractor.ID_OVERFLOW % [e.rank for e in exs if e.id_overflow()])
    for r in range(world - 1):  # phase C
        exs[r].records_recv_buffer().copy_(exs[r + 1].records_send_buffer())
    return [e.phase_faces(counts) for e in exs]  # phase D
