"""Framebuffer row-strip tiling across the GPUs of one node (BASELINE config 4).

The scene is replicated; the WxH framebuffer is cut into `world` contiguous row strips.  Per frame a
rank (one process per GPU, torch.distributed over RCCL/xGMI) does

    G-buffer for its rows                  (only its own: on a 1/8 strip of a 1080p frame re-rendering +-5 halo rows in 8-row
                                            tiles was +15 % G-buffer work)
    phase A on its rows                    (primary hit, RIS, shadow ray, temporal merge, publish)
    exchange HALO rows of published reservoirs AND of the G-buffer id / normal / depth planes with the strip above and
                                            below   <- point-to-point send/recv, 48 + 20 B/px: 653 KB per edge at 1080p
    phase B on its rows                    (spatial reuse, shade, accumulate): the interior rows while the
                                            halo is in flight, the two 5-row border bands after it arrived

The spatial taps reach y-4..y+5 (src/restir.cu:49-56), so HALO = 5 rows is exact.

Temporal reuse reads last frame's reservoirs and G-buffer history at devMotion[idx] (restir.cu:20-45).
With the reference's default static camera that is the pixel itself, i.e. strip-local.  When the
camera moves the reprojected pixel may belong to another strip, so `share_history=True` all-gathers,
after each frame, the rows every rank just produced of (a) the reservoirs the next temporal merge
reads and (b) the G-buffer id/normal/depth planes that become "last" -- 60 B/px, exact for any motion.

The driver is backend-agnostic: `HipBackend` runs librestir_hip; tests/ provide an oracle-backed
double so the decomposition and the exchanges are exercised with gloo on CPU.
"""

HALO = 5


def strip_bounds(height, world, rank):
    """Contiguous row strips whose heights differ by at most one row."""
    base, rem = divmod(height, world)
    y0 = rank * base + min(rank, rem)
    return y0, y0 + base + (1 if rank < rem else 0)


def rebalance_bounds(bounds, times, height, quantum=8, min_rows=8):
    """Cost-balanced strips (rows differ in cost, and the slowest strip sets the frame time).

    bounds: current [(y0, y1)] per rank, times: measured cost of each strip (any unit).  The cost per row is taken
    as constant inside a strip; new boundaries are placed at equal shares of the cumulative cost, rounded to
    `quantum` rows (the kernels work in 8-row tiles) and kept at least `min_rows` (>= HALO) tall.  Pure function of
    its arguments, so every rank derives the same bounds from the same all-gathered times."""
    world = len(bounds)
    if world == 1:
        return [(0, height)]
    row_cost = []
    for (a, b), t in zip(bounds, times):
        row_cost += [float(t) / (b - a)] * (b - a)
    total = sum(row_cost)
    cuts, acc, y = [], 0.0, 0
    for k in range(1, world):
        target = total * k / world
        while y < height and acc + row_cost[y] <= target:
            acc += row_cost[y]; y += 1
        cuts.append(y)
    lo = max(min_rows, quantum)
    out, prev = [], 0
    for k, c in enumerate(cuts):
        c = int(round(c / quantum)) * quantum
        c = max(c, prev + lo)                                   # tall enough
        c = min(c, height - lo * (world - 1 - k))               # leave room for the strips below
        out.append((prev, c)); prev = c
    out.append((prev, height))
    return out


class StripRenderer:
    """runCuda (src/main.cpp:146-185) for one rank of a row-strip decomposition.

    backend must provide: gbuffer_render(y0,y1), phase_a(looper,reuse,y0,y1), phase_b(iter,reuse,y0,y1),
    end_frame(), halo_pack(y0,rows)->tensor, halo_unpack(y0,rows,tensor), empty(nbytes)->tensor,
    history_bytes(rows), history_pack(y0,rows)->tensor, history_unpack(y0,rows,tensor).
    """

    def __init__(self, backend, world, rank, height, dist=None, share_history=False, bounds=None):
        self.b = backend
        self.world, self.rank, self.height = world, rank, height
        self.dist = dist
        self.share_history = share_history and world > 1
        # bounds: optional explicit [(y0, y1)] per rank (cost-balanced strips); must tile [0, height) in rank order
        self.bounds = list(bounds) if bounds is not None else [strip_bounds(height, world, r) for r in range(world)]
        if len(self.bounds) != world or self.bounds[0][0] != 0 or self.bounds[-1][1] != height or \
                any(self.bounds[i][1] != self.bounds[i + 1][0] for i in range(world - 1)):
            raise ValueError("strip bounds must tile the framebuffer rows in rank order")
        self.y0, self.y1 = self.bounds[rank]
        if world > 1 and min(b[1] - b[0] for b in self.bounds) < HALO:
            raise ValueError("strips must be at least HALO rows tall")
        self.gy0, self.gy1 = self.y0, self.y1            # G-buffer rows rendered here (the halo rows' planes arrive with the halo)
        self.up = rank - 1 if rank > 0 else None
        self.down = rank + 1 if rank + 1 < world else None
        self.max_rows = max(b[1] - b[0] for b in self.bounds)
        self.looper = 0

    def _host_visible(self):
        """RCCL orders its transfers after the work already enqueued on the current stream.  gloo -- the stand-in used to rehearse
        the multi-process path on a one-GPU box -- reads device buffers from the host without any stream ordering: wait for the
        packing copies first when launches are asynchronous."""
        if self.dist is not None and self.dist.get_backend() == "gloo" and hasattr(self.b, "torch") and self.b.torch.cuda.is_available():
            self.b.torch.cuda.synchronize()

    def start_halo_exchange(self):
        """Pack the strip's border rows and post the sends / receives; returns what finish_halo_exchange needs."""
        d = self.dist
        ops, recvs = [], []
        if self.up is not None:
            send = self.b.halo_pack(self.y0, HALO)
            recv = self.b.empty(send.numel())
            ops += [d.P2POp(d.isend, send, self.up), d.P2POp(d.irecv, recv, self.up)]
            recvs.append((self.y0 - HALO, recv))
        if self.down is not None:
            send = self.b.halo_pack(self.y1 - HALO, HALO)
            recv = self.b.empty(send.numel())
            ops += [d.P2POp(d.isend, send, self.down), d.P2POp(d.irecv, recv, self.down)]
            recvs.append((self.y1, recv))
        if ops:
            self._host_visible()
        works = d.batch_isend_irecv(ops) if ops else []
        return works, recvs, ops                 # ops keeps the send buffers alive until the wait

    def finish_halo_exchange(self, pending):
        works, recvs, _ = pending
        for w in works:
            w.wait()
        for y, buf in recvs:
            self.b.halo_unpack(y, HALO, buf)

    def exchange_halo(self):
        self.finish_halo_exchange(self.start_halo_exchange())

    def phase_b_overlapped(self, iteration, reuse):
        """Phase B with the halo exchange in flight: the interior rows (whose +-5-row taps stay inside the strip)
        run while the border rows of the neighbours travel; the two border bands follow once they have arrived."""
        b = self.b
        pending = self.start_halo_exchange()
        top_end = min(self.y0 + HALO, self.y1) if self.up is not None else self.y0
        bot_start = max(self.y1 - HALO, top_end) if self.down is not None else self.y1
        if bot_start > top_end:
            b.phase_b(iteration, reuse, top_end, bot_start)
        self.finish_halo_exchange(pending)
        if top_end > self.y0:
            b.phase_b(iteration, reuse, self.y0, top_end)
        if self.y1 > bot_start:
            b.phase_b(iteration, reuse, bot_start, self.y1)

    # ---- LeveledEAWFilter on strips (BASELINE config 5) ----------------------------------------------------------------
    def _exchange_rows(self, rows, get, put):
        """Send this strip's first / last `rows` rows to the strip above / below and receive theirs into the rows just outside
        the strip.  get(y, rows) -> contiguous tensor, put(y, rows, tensor)."""
        d = self.dist
        ops, recvs = [], []
        for peer, send_y, recv_y in ((self.up, self.y0, self.y0 - rows), (self.down, self.y1 - rows, self.y1)):
            if peer is None:
                continue
            send = get(send_y, rows)
            recv = self.b.torch.empty_like(send)
            ops += [d.P2POp(d.isend, send, peer), d.P2POp(d.irecv, recv, peer)]
            recvs.append((recv_y, recv))
        if ops:
            self._host_visible()
            for w in d.batch_isend_irecv(ops):
                w.wait()
        for y, buf in recvs:
            put(y, rows, buf)

    def eaw_filter(self):
        """LeveledEAWFilter::filter (src/denoiser.cu:453-477) on this strip's rows of the radiance image: five a-trous levels
        whose taps reach 2 << level rows beyond the strip.  The G-buffer rows the taps look at (32 at most) come from the
        neighbouring strips once, and before each level the strips swap the 2 << level border rows of that level's input --
        the same values a full-frame filter reads there, so the result equals the full-frame filter's rows bit for bit.
        Returns the backend's result buffer (rows [y0, y1) valid)."""
        b = self.b
        reach = 2 << 4
        if self.world > 1:
            if min(y1 - y0 for y0, y1 in self.bounds) < reach:
                raise ValueError("EAW on strips needs strips of at least %d rows" % reach)
            self._exchange_rows(reach, b.gbuffer_rows_get, b.gbuffer_rows_put)
        b.eaw_positions(max(0, self.y0 - reach), min(self.height, self.y1 + reach))
        for level in range(5):
            if self.world > 1:
                self._exchange_rows(2 << level, lambda y, n: b.eaw_rows_get(level, y, n), lambda y, n, t: b.eaw_rows_put(level, y, n, t))
            b.eaw_level(level, self.y0, self.y1)
        return b.eaw_result()

    def exchange_history(self):
        """All-gather of the rows this frame produced that the next frame's temporal merge may read."""
        d = self.dist
        rows = self.y1 - self.y0
        mine = self.b.history_pack(self.y0, rows)
        nmax = self.b.history_bytes(self.max_rows)
        send = self.b.empty(nmax)
        send[: mine.numel()] = mine
        out = [self.b.empty(nmax) for _ in range(self.world)]
        self._host_visible()
        d.all_gather(out, send)
        for r, (a, bnd) in enumerate(self.bounds):
            if r != self.rank:
                self.b.history_unpack(a, bnd - a, out[r][: self.b.history_bytes(bnd - a)])

    def frame(self, reuse, iteration=0, denoise=False):
        """One runCuda frame; denoise=True also runs LeveledEAWFilter on the strip before GBuffer::update (the reference's
        order, src/main.cpp:146-185) and leaves its result in self.filtered."""
        b = self.b
        if self.world == 1:
            b.gbuffer_render(0, self.height)
            b.phase_a(self.looper, reuse, 0, self.height)
            b.phase_b(iteration, reuse, 0, self.height)
        else:
            b.gbuffer_render(self.gy0, self.gy1)
            b.phase_a(self.looper, reuse, self.y0, self.y1)
            if reuse & 2:
                self.phase_b_overlapped(iteration, reuse)
            else:
                b.phase_b(iteration, reuse, self.y0, self.y1)
        if denoise:
            self.filtered = self.eaw_filter()
        b.end_frame()
        if self.share_history and (reuse & 1):
            self.exchange_history()
        self.looper += 1


def calibrate_bounds(backend, world, rank, height, dist, synchronize, reuse=3, rounds=3, frames=8, denoise=False, min_rows=8):
    """Cost-balanced strip bounds from measurement: every rank times its own strip's kernels alone (no halo exchange, so a
    slow neighbour does not leak into the number), the times are all-gathered, `rebalance_bounds` moves the boundaries, and the
    loop repeats.  Rows around the horizon of a scene cost several times what sky or floor rows cost, and the slowest strip
    sets the frame time.  All ranks compute the same bounds (pure function of the gathered times).

    denoise: the frames include LeveledEAWFilter on the strip (its cost per row is uniform, which flattens the balance);
    min_rows: the shortest strip allowed (32 with the filter, whose levels reach 32 rows into the neighbours).

    `synchronize()` waits for the backend's device work (a no-op for a CPU backend).  The frames rendered here leave
    reservoir and G-buffer contents behind that a fresh run would not have (stale slots are visible to the spatial pass, Q1):
    the caller renders for real with a NEW backend (fresh buffers)."""
    import time
    bounds = [strip_bounds(height, world, r) for r in range(world)]
    if world == 1:
        return bounds
    torch = backend.torch
    device = "cpu" if dist.get_backend() == "gloo" else backend.empty(1).device      # timings travel over the control plane
    for _ in range(rounds):
        s = StripRenderer(backend, world, rank, height, bounds=bounds)
        s.start_halo_exchange = lambda: ([], [], [])            # the strip's own kernels only
        s._exchange_rows = lambda rows, get, put: None
        for _ in range(2):
            s.frame(reuse, 0, denoise=denoise)
        synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            s.frame(reuse, 0, denoise=denoise)
        synchronize()
        mine = torch.tensor([(time.perf_counter() - t0) / frames], dtype=torch.float64, device=device)
        out = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(out, mine)
        bounds = rebalance_bounds(bounds, [float(t[0]) for t in out], height, min_rows=min_rows)
    return bounds


class HipBackend:
    """librestir_hip objects of one rank."""

    def __init__(self, capi, scene, cam, width, height):
        import torch
        self.torch = torch
        self.capi = capi
        self.scene, self.cam = scene, cam
        self.W, self.H = width, height
        # c10d orders a collective after the work on torch's CURRENT stream, and the buffers it sends are packed by copies the
        # library enqueues on ITS stream: they must be the same stream (by default both are the legacy default stream)
        capi.set_stream(torch.cuda.current_stream().cuda_stream)
        self.gbuf = capi.GBuffer(width, height)
        self.restir = capi.ReSTIR(width, height)
        self.image = torch.zeros((width * height, 3), dtype=torch.float32, device="cuda")

    def empty(self, nbytes):
        return self.torch.empty(nbytes, dtype=self.torch.uint8, device="cuda")

    def gbuffer_render(self, y0, y1):
        self.gbuf.render(self.scene, self.cam, y0, y1)

    def phase_a(self, looper, reuse, y0, y1):
        self.restir.phase_a(self.scene, self.cam, self.gbuf, looper, reuse, y0, y1)

    def phase_b(self, iteration, reuse, y0, y1):
        self.restir.phase_b(self.scene, self.cam, self.gbuf, self.image.data_ptr(), iteration, reuse, y0, y1)

    def end_frame(self):
        self.restir.end_frame()
        self.gbuf.update(self.cam)

    # halo = published reservoirs (48 B/px) + the G-buffer id / normal / depth rows the spatial taps compare against (20 B/px)
    def halo_pack(self, y0, rows):
        nr = self.restir.halo_bytes(rows)
        buf = self.empty(nr + self.gbuf.rows_bytes(rows))
        self.restir.halo_pack(y0, rows, buf.data_ptr())
        self.gbuf.rows_pack(0, y0, rows, buf.data_ptr() + nr)
        return buf

    # LeveledEAWFilter on strips: level l reads input l (the radiance image for l = 0, else the output of level l - 1) and
    # writes one of two full-frame buffers, alternating, so that the result of the five levels ends in buffer 0
    def _eaw_init(self):
        if getattr(self, "eaw", None) is None:
            self.eaw = self.capi.EAWFilter(self.W, self.H, 5)
            self.eaw_buf = [self.torch.zeros_like(self.image), self.torch.zeros_like(self.image)]

    def _eaw_input(self, level):
        self._eaw_init()
        return self.image if level == 0 else self.eaw_buf[(level - 1) % 2]

    def gbuffer_rows_get(self, y, rows):
        buf = self.empty(self.gbuf.rows_bytes(rows))
        self.gbuf.rows_pack(0, y, rows, buf.data_ptr())
        return buf

    def gbuffer_rows_put(self, y, rows, buf):
        self.gbuf.rows_unpack(0, y, rows, buf.contiguous().data_ptr())

    def eaw_rows_get(self, level, y, rows):
        return self._eaw_input(level)[y * self.W:(y + rows) * self.W]          # rows of a row-major image are contiguous

    def eaw_rows_put(self, level, y, rows, buf):
        self._eaw_input(level)[y * self.W:(y + rows) * self.W].copy_(buf)

    def eaw_positions(self, y0, y1):
        self._eaw_init()
        self.eaw.positions_rows(self.gbuf, self.cam, y0, y1)

    def eaw_level(self, level, y0, y1):
        self.eaw.level_rows(self.eaw_buf[level % 2].data_ptr(), self._eaw_input(level).data_ptr(), self.gbuf, level, y0, y1)

    def eaw_result(self):
        return self.eaw_buf[0]

    def halo_unpack(self, y0, rows, buf):
        self.restir.halo_unpack(y0, rows, buf.data_ptr())
        self.gbuf.rows_unpack(0, y0, rows, buf.data_ptr() + self.restir.halo_bytes(rows))

    # history = reservoirs the next temporal merge reads (buffer 1 after end_frame) + "last" G-buffer planes
    def history_bytes(self, rows):
        return self.restir.rows_bytes(1, rows) + self.gbuf.rows_bytes(rows)

    def history_pack(self, y0, rows):
        nr = self.restir.rows_bytes(1, rows)
        buf = self.empty(self.history_bytes(rows))
        self.restir.rows_pack(1, y0, rows, buf.data_ptr())
        self.gbuf.rows_pack(1, y0, rows, buf.data_ptr() + nr)
        return buf

    def history_unpack(self, y0, rows, buf):
        nr = self.restir.rows_bytes(1, rows)
        buf = buf.contiguous()
        self.restir.rows_unpack(1, y0, rows, buf.data_ptr())
        self.gbuf.rows_unpack(1, y0, rows, buf.data_ptr() + nr)
