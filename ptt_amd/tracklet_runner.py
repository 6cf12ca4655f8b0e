"""N4 — the reference's sequential tracking loop (TrackingEvaluator.test_batch / test_frame,
tools/eval_utils/eval_tracking_utils.py:77-152) with everything but a few float64 box updates on the device.

"Tracklet frames/sec" in the reference is this loop at B = 1: for frame i, crop the cloud around the PREVIOUS result
box and resample it to 1024 points (prepare_search :155-184), build the template from the first and the previous
frame's crops resampled to 512 (prepare_template :186-229), run the model (:231-264), take the best proposal and move
the box (post_process :266-274). Frame i needs frame i-1's box, so a tracklet is strictly sequential; different
tracklets are independent.

TrackletRunner keeps the clouds of up to `batch` tracklets resident in HBM and advances them in LOCKSTEP, one frame of
every tracklet per step:

    host    crop bounds of B boxes (float64, ptt_track_crop_bounds)       -> one job table in pinned memory, one upload
    device  ptt_crop_compact_f32   2B jobs: search crop, previous-frame template crop     (1 launch)
            ptt_regularize_f32     2B jobs: resample to 1024 / 512 straight into the model's input buffers (1 launch)
            the tracker forward for the B frames                                           (hipGraph replay)
            ptt_select_box_f32     best proposal of each frame                             (inside the graph)
    host    one (B,5) + (B,2,2) read-back, float64 box update of B boxes (ptt_track_box_by_offset)

At batch = 1 this is the reference's own mode (one tracklet, one frame at a time) with the per-frame PCIe traffic cut
to ~0.5 KB; at batch = 48 it is the throughput mode for evaluating a dataset's tracklets.

REF_BOX = previous_result and SHAPE_AGGREGATION = firstandprevious (the shipped tools/cfgs/*/ptt.yaml:149-150) are what
is implemented.
"""
import numpy as np
import torch

from . import graph_policy, ops
from .hot_path import GraphedHotPath, TrackerThroughput


class _BoxedForward(object):
    """tracker forward + best-proposal selection as ONE capturable callable: (search, template) -> (B,5) rows, the best
    proposal of each frame picked on the device (ptt_select_box_f32) — or, select=False, all (B,P,5) proposals: at a handful of
    tracklets the host takes the arg-max itself from the one read-back it makes anyway (np.argmax, the reference's own
    post_process, eval_tracking_utils.py:267-269): one launch less on a chain where launches are what a frame costs."""

    def __init__(self, tracker, select=True):
        self.fwd = TrackerThroughput(tracker)
        self.select = select

    def __call__(self, search, template):
        boxes = self.fwd(search, template)['pred_box_data'].contiguous()
        return ops.select_box(boxes) if self.select else boxes


class TrackletRunner(object):
    def __init__(self, tracker, device, batch=1, search_size=1024, template_size=512, search_offset=0.0,
                 search_scale=1.25, model_offset=0.0, model_scale=1.25, use_z=True, use_graph=True):
        """`tracker`: ptt_amd.models.trackers.PTT in eval mode on `device`. Sizes / offsets / scales are
        DATA_CONFIG.{SEARCH,TEMPLATE}_INPUT_SIZE, SEARCH_BB_*, MODEL_BB_* and USE_Z_AXIS
        (tools/cfgs/kitti_models/ptt.yaml:8-17)."""
        self.tracker = tracker
        self.device = torch.device(device)
        self.B = int(batch)
        self.S, self.T = int(search_size), int(template_size)
        self.search_offset, self.search_scale = float(search_offset), float(search_scale)
        self.model_offset, self.model_scale = float(model_offset), float(model_scale)
        self.use_z = bool(use_z)
        self.use_graph = use_graph
        dev, B = self.device, self.B
        self.search = torch.zeros((B, self.S, 3), dtype=torch.float32, device=dev)
        self.template = torch.zeros((B, self.T, 3), dtype=torch.float32, device=dev)
        self.counts = torch.zeros((B, 3), dtype=torch.int32, device=dev)          # search, first-frame, previous-frame
        self.info = torch.zeros((B, 2, 2), dtype=torch.int32, device=dev)         # (n, draws used) of search / template
        self.crop_jobs_dev = torch.zeros(2 * B * ops.CROP_JOB.itemsize, dtype=torch.uint8, device=dev)
        self.crop_jobs_host = torch.zeros(2 * B * ops.CROP_JOB.itemsize, dtype=torch.uint8).pin_memory()   # staging: the
        self.crop_jobs_host_np = self.crop_jobs_host.numpy().view(ops.CROP_JOB)   # host waits for every step's result
        #                                                                           before it rewrites the table
        self.reg_jobs_dev = torch.zeros(2 * B * ops.REGULARIZE_JOB.itemsize, dtype=torch.uint8, device=dev)
        self.draws = ops.mt19937_draws(dev, max(8192, 4 * max(self.S, self.T) + 1024))
        # a handful of tracklets: the crop table rides in the crop launch's arguments (no upload), and the host picks the best
        # proposal itself from the (B,P,5) read-back — two launches less per frame of a chain that is launches
        self.few = 2 * B <= ops.CROP_JOBS_BY_VALUE_MAX
        if self.few and use_graph:
            # one read-back buffer: (B,P,5) proposals, then the (B,2,2) resampling counts
            self.P = int(tracker.box_voting_head.model_cfg.SA_CONFIG.NPOINTS)
            self.n_box = B * self.P * 5
            self.readback = torch.zeros(self.n_box + B * 4, dtype=torch.float32, device=dev)
            self.readback_host = torch.zeros(self.n_box + B * 4, dtype=torch.float32).pin_memory()
            self.info = self.readback[self.n_box:].view(torch.int32).view(B, 2, 2)
        self.result_host = None if self.few else torch.empty((B, 5), dtype=torch.float32).pin_memory()
        self.info_host = torch.empty((B, 2, 2), dtype=torch.int32).pin_memory()
        self._model = _BoxedForward(tracker, select=not self.few)
        self._graph = None
        self._frame = None                                           # few tracklets: crops + resampling + model + read-backs, one graph
        self._done = torch.cuda.Event()
        self.stream = None                                           # run_overlapped gives every runner its own stream
        self.profile = None       # set to {} before run(): per-frame host_pre / device / host_post milliseconds are appended

    # ------------------------------------------------------------------ device buffers of one group of tracklets
    def _load(self, tracklets):
        """tracklets: list (<= batch) of (clouds, boxes): clouds = list of (3,N_i) float32 arrays / tensors, boxes =
        list of (center (3), wlh (3), quat (4)) ground-truth boxes (frame 0 initialises; wlh[1] enters the search crop,
        eval_tracking_utils.py:165-169)."""
        dev, B = self.device, self.B
        self._ensure_graph()
        T = max(len(c) for c, _ in tracklets)
        self.clouds = []
        self.ptr = np.zeros((T, B), np.uint64)
        self.ld = np.zeros((T, B), np.int64)
        self.npts = np.zeros((T, B), np.int32)
        cap = 1
        for b, (clouds, _) in enumerate(tracklets):
            if all(isinstance(c, np.ndarray) for c in clouds) and len(clouds) > 0:
                # host arrays: ONE upload per tracklet — the frames side by side in a (3, sum N_i) buffer, every frame a
                # column range of it (the crop kernel takes the row stride) — instead of one small copy per frame
                sizes = [c.shape[1] for c in clouds]
                packed = torch.from_numpy(np.ascontiguousarray(np.concatenate([c[0:3] for c in clouds], axis=1), np.float32))
                packed = packed.to(dev)                  # pageable copy: pinning a fresh buffer per tracklet costs more than it saves
                offs = np.concatenate([[0], np.cumsum(sizes)])
                row = [packed[:, offs[i]:offs[i + 1]] for i in range(len(clouds))]
            else:
                row = [(c if isinstance(c, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(c, np.float32)))
                       .to(dev, torch.float32) for c in clouds]
                row = [t if t.stride(1) == 1 else t.contiguous() for t in row]
            for i, t in enumerate(row):
                self.ptr[i, b], self.ld[i, b], self.npts[i, b] = t.data_ptr(), t.stride(0), t.shape[1]
                cap = max(cap, t.shape[1])
            self.clouds.append(row)
        self.cap = cap
        self.crop_out = torch.zeros((B, 3, cap, 3), dtype=torch.float32, device=dev)   # slot 0 search, 1 first, 2 previous
        esz = self.crop_out.element_size()
        self.out_ptr = (self.crop_out.data_ptr() + (np.arange(B)[:, None] * 3 + np.arange(3)[None]) * (cap * 3 * esz)).astype(np.uint64)
        self.cnt_ptr = (self.counts.data_ptr() + (np.arange(B)[:, None] * 3 + np.arange(3)[None]) * 4).astype(np.uint64)
        self.ptr[self.npts == 0] = self.crop_out.data_ptr()        # empty jobs still carry a valid address
        # per tracked frame i the cloud fields of the interleaved 2B-job table: even jobs = frame i, odd jobs = frame i - 1
        ff = [np.zeros((T, 2 * B), dt) for dt in (np.uint64, np.int64, np.int32)]
        for dst, src in zip(ff, (self.ptr, self.ld, self.npts)):
            dst[:, 0::2] = src
            dst[1:, 1::2] = src[:-1]
        self.frame_fields = ff
        # the resampling jobs never change within a group: fixed segment / output pointers
        rj = np.zeros(2 * B, ops.REGULARIZE_JOB)
        s, t = rj[0::2], rj[1::2]
        s['seg'][:, 0], s['seg_count'][:, 0], s['seg_capacity'][:, 0] = self.out_ptr[:, 0], self.cnt_ptr[:, 0], cap
        s['n_seg'], s['input_size'] = 1, self.S
        s['out'] = self.search.data_ptr() + np.arange(B) * (self.S * 3 * 4)
        s['info'] = self.info.data_ptr() + np.arange(B) * 16
        for k, slot in enumerate((1, 2)):                          # get_model([PC_0, PC_{i-1}], ...) order (:189-194)
            t['seg'][:, k], t['seg_count'][:, k], t['seg_capacity'][:, k] = self.out_ptr[:, slot], self.cnt_ptr[:, slot], cap
        t['n_seg'], t['input_size'] = 2, self.T
        t['out'] = self.template.data_ptr() + np.arange(B) * (self.T * 3 * 4)
        t['info'] = self.info.data_ptr() + np.arange(B) * 16 + 8
        ops.upload_jobs(rj, self.reg_jobs_dev)

    def _crop_jobs(self, frame_a, slot_a, cfg_a, frame_b, slot_b, cfg_b, launch=True):
        """The 2B-entry crop table of one step, built in the pinned staging buffer and copied to the device: job 2b =
        cloud `frame_a` of tracklet b cropped around the tracklet's current box into slot_a, job 2b+1 likewise (frame
        None / past the tracklet's end: an empty job -> count 0 -> an all-zero resampled cloud). cfg = (offset, scale,
        extra2 | None); the float64 bounds come from ptt_track_crop_bounds (microseconds for 48 boxes)."""
        jobs = self.crop_jobs_host_np
        for k, (frame, slot, cfg) in enumerate(((frame_a, slot_a, cfg_a), (frame_b, slot_b, cfg_b))):
            half = jobs[k::2]
            half['out'], half['count'] = self.out_ptr[:, slot], self.cnt_ptr[:, slot]
            if frame is None or frame >= self.ptr.shape[0]:
                half['points'], half['n_points'] = self.crop_out.data_ptr(), 0
                continue
            half['points'], half['ld'], half['n_points'] = self.ptr[frame], self.ld[frame], self.npts[frame]
            ops.track_crop_bounds(self.boxes, cfg[0], cfg[1], cfg[2], half, job_stride=2)
        if self.few and self.use_graph and launch is False:
            return                                   # the frame graph reads the pinned table itself
        if self.few:
            ops.crop_compact_host(jobs, 2 * self.B, self.device)
        else:
            self.crop_jobs_dev.copy_(self.crop_jobs_host, non_blocking=True)
            ops.crop_compact(self.crop_jobs_dev, 2 * self.B)

    def _frame_jobs(self, i, extra2):
        """The crop table of tracked frame i >= 1 written into the pinned staging buffer with three array assignments and two
        calls: job 2b = cloud i of tracklet b around its current box into slot 0 (the search crop, `extra2` = gt_wlh1 * 0.6),
        job 2b + 1 = cloud i - 1 into slot 2 (get_model's previous-frame segment). The per-frame cloud fields were laid out for
        all frames by _load (self.frame_fields); `out` / `count` were set for these slots once (_steps)."""
        jobs = self.crop_jobs_host_np
        pts, ld, npts = self.frame_fields
        jobs['points'], jobs['ld'], jobs['n_points'] = pts[i], ld[i], npts[i]
        ops.track_crop_bounds(self.boxes, self.search_offset, self.search_scale, extra2, jobs[0::2], job_stride=2)
        ops.track_crop_bounds(self.boxes, self.model_offset, self.model_scale, None, jobs[1::2], job_stride=2)

    def _launch_jobs(self):
        jobs = self.crop_jobs_host_np
        if self.few:
            ops.crop_compact_host(jobs, 2 * self.B, self.device)
        else:
            self.crop_jobs_dev.copy_(self.crop_jobs_host, non_blocking=True)
            ops.crop_compact(self.crop_jobs_dev, 2 * self.B)

    # ------------------------------------------------------------------ one group in lockstep
    def _steps(self, tracklets):
        """Generator form of one lockstep group: every `yield` sits between "frame i's device work is enqueued" and
        "the host waits for it", so that a driver can enqueue ANOTHER group's frame in between (run_overlapped). The
        generator's return value (StopIteration.value) is the group's result list."""
        B = self.B
        n = len(tracklets)
        self._load(tracklets)
        lengths = np.array([len(c) for c, _ in tracklets] + [0] * (B - n))
        T = int(lengths.max())
        boxes = self.boxes = np.zeros(B, ops.TRACK_BOX)
        boxes['wlh'], boxes['quat'][:, 0] = 1.0, 1.0
        gt_wlh1 = np.zeros((T, B))
        for b, (_, gts) in enumerate(tracklets):
            boxes['center'][b], boxes['wlh'][b], boxes['quat'][b] = gts[0][0], gts[0][1], gts[0][2]
            for i, bx in enumerate(gts):
                gt_wlh1[i, b] = bx[1][1]
        self.crop_jobs_host_np['capacity'] = self.cap
        wlh0 = boxes['wlh'].copy()
        history = [(np.ones(B, np.int32), boxes['center'].copy(), boxes['quat'].copy(), None)]   # per step: whole-batch copies
        rng_pos = np.zeros(B, np.int64)            # where numpy's global generator stands for each tracklet
        model_cfg = (self.model_offset, self.model_scale, None)

        # frame 0: the first-frame template crop (get_model's first segment) is fixed for the whole tracklet
        self._crop_jobs(0, 1, model_cfg, None, 2, model_cfg)
        # the job table travels through ONE pinned staging buffer: its copy must have left the host before frame 1's
        # table is written into it (every later frame waits for its boxes anyway)
        self._done.record(torch.cuda.current_stream(self.device))
        self._done.synchronize()
        # every tracked frame crops into the same slots: search -> 0, previous-frame template segment -> 2
        jobs = self.crop_jobs_host_np
        jobs['out'][0::2], jobs['count'][0::2] = self.out_ptr[:, 0], self.cnt_ptr[:, 0]
        jobs['out'][1::2], jobs['count'][1::2] = self.out_ptr[:, 2], self.cnt_ptr[:, 2]
        extra_search = np.ascontiguousarray(gt_wlh1 * 0.6)        # (T, B): gt_box.wlh[1] * 0.6 enters the search crop (:321)
        est_buf = np.zeros((B, 5), np.float32)
        active_all = (np.arange(T)[:, None] < lengths[None, :]).astype(np.int32)
        views = None                                             # numpy views of the pinned read-back buffers, made once

        prof = self.profile
        if prof is not None:
            import time
            for k in ('host_pre_ms', 'device_ms', 'host_post_ms', 'frame_ms'):
                prof.setdefault(k, [])
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(1, T):
            if prof is not None:
                t_a = time.perf_counter()
                ev0.record(torch.cuda.current_stream(self.device))
            active = active_all[i]
            # both crops of frame i are taken around the previous RESULT box (prepare_search :156-157, prepare_template
            # :189-194 with results_BBs[frame_id - 1]); a finished tracklet's later frames have n_points 0
            if self.few and self.use_graph:
                # a handful of tracklets: the WHOLE frame is one hipGraph replay — crops (their table read from pinned host
                # memory, rewritten here), resampling, read-back of the draw counts, tracker, read-back of the proposals
                self._frame_jobs(i, extra_search[i])
                if self._frame is None:
                    self._capture_frame()
                self._frame.replay()
            else:
                self._frame_jobs(i, extra_search[i])
                self._launch_jobs()
                ops.regularize(self.reg_jobs_dev, 2 * B, self.draws)
                self.info_host.copy_(self.info, non_blocking=True)   # behind the resampling, ahead of the model: off the frame's tail
                rows = self._forward()
                if self.result_host is None or self.result_host.shape != rows.shape:
                    self.result_host = torch.empty(tuple(rows.shape), dtype=torch.float32).pin_memory()
                self.result_host.copy_(rows, non_blocking=True)
            self._done.record(torch.cuda.current_stream(self.device))
            if prof is not None:
                ev1.record(torch.cuda.current_stream(self.device))
                t_b = time.perf_counter()
            yield i
            self._done.synchronize()
            if prof is not None:
                t_c = time.perf_counter()
            if self.few and self.use_graph:                      # one buffer: (B,P,5) proposals, then the resampling counts
                if views is None:
                    host = self.readback_host.numpy()
                    views = (host[:self.n_box].reshape(B, self.P, 5), host[self.n_box:].view(np.int32).reshape(B, 2, 2))
            elif views is None or views[2] is not self.result_host:
                # (B,5) float32: x, y, z, theta (degrees), score
                views = (self.result_host.numpy(), self.info_host.numpy(), self.result_host)
            est, info = views[0], views[1]
            # post_process (:266-274) in one call (ptt_track_select_update): the first arg-max of the scores where the read-back is
            # (B,P,5) (:267-269), box_i = get_box_by_offset(box_{i-1}, best proposal, USE_Z_AXIS). An implausibly large x / y offset
            # is redrawn from numpy's GLOBAL generator (:205-208), whose state then is "seeded with 1 and advanced by the template's
            # (else the search's) resampling draws" — the draw counts come back with the boxes; a draw table that ran out raises
            ops.track_select_update(est, info, boxes, self.use_z, active, rng_pos, est_buf)
            history.append((active, boxes['center'].copy(), boxes['quat'].copy(), est_buf[:, 4].copy()))
            if prof is not None:
                t_d = time.perf_counter()
                ev1.synchronize()
                prof['host_pre_ms'].append((t_b - t_a) * 1e3)        # crop bounds, job upload, enqueue of the frame's launches
                prof['device_ms'].append(ev0.elapsed_time(ev1))      # crop + resample + model graph + read-back, on the device
                prof['host_post_ms'].append((t_d - t_c) * 1e3)       # float64 box update, history
                prof['frame_ms'].append((t_d - t_a) * 1e3)
        # per-tracklet result lists, assembled once (three array copies per step instead of 3 x B small ones)
        results = []
        for b in range(n):
            rows = []
            for act, c, q, sc in history:
                if act[b]:
                    rows.append((c[b], wlh0[b], q[b]) if sc is None else (c[b], wlh0[b], q[b], float(sc[b])))
            results.append(rows)
        return results

    def _run_group(self, tracklets):
        gen = self._steps(tracklets)
        while True:
            try:
                next(gen)
            except StopIteration as stop:
                return stop.value

    def _frame_body(self):
        ops.crop_regularize_pinned(self.crop_jobs_host, self.reg_jobs_dev, 2 * self.B, self.draws)     # crop w, then resampling w
        rows = self._model(self.search, self.template)
        if rows.data_ptr() != self.readback.data_ptr():      # a box head that did not take the preallocated output: one copy
            self.readback[:self.n_box].view(self.B, self.P, 5).copy_(rows)
        return rows

    def _capture_frame(self):
        """One frame of a handful of tracklets as ONE hipGraph: called at the first tracked frame of the first group, when the
        job tables hold valid pointers (the warm-up runs execute them). The box head writes its proposals straight into the
        read-back buffer whose tail the resampling kernel fills with its draw counts (`self.info` lives there): ONE device-to-
        host copy per frame."""
        head = self.tracker.box_voting_head
        with torch.no_grad():
            cur = torch.cuda.current_stream(self.device)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(cur)
            head.pred_box_out = self.readback[:self.n_box].view(self.B, self.P, 5)
            try:
                with torch.cuda.stream(side):
                    for _ in range(3):                       # weight packing / LDS attributes happen here, not in capture
                        rows = self._frame_body()
                cur.wait_stream(side)
                torch.cuda.synchronize(self.device)
                self._frame = torch.cuda.CUDAGraph()
                with graph_policy.capture_scope(), torch.cuda.graph(self._frame):
                    self._frame_body()
                    self.readback_host.copy_(self.readback, non_blocking=True)
            finally:
                head.pred_box_out = None                     # the tracker may be shared: the hook is only live during capture

    def _ensure_graph(self):
        """The tracker forward + box selection for B frames as a hipGraph whose static inputs ARE the buffers the
        resampling kernel writes (no staging copy)."""
        if self.few:
            return                                           # the frame graph (_capture_frame) holds the model
        if self.use_graph and self._graph is None:
            with torch.no_grad():
                self._graph = GraphedHotPath(self._model, self.search, self.template)
            self.search, self.template = self._graph.search, self._graph.template

    def _forward(self):
        with torch.no_grad():
            if not self.use_graph:
                return self._model(self.search, self.template)
            return self._graph()                                  # inputs are already in the graph's static buffers

    # ------------------------------------------------------------------ public
    def run(self, tracklets):
        """tracklets: list of (clouds, boxes) as in `_load`. Returns, per tracklet, the list of result boxes
        [(center, wlh, quat[, score]), ...] — element 0 is the frame-0 ground-truth box, as in the reference
        (eval_tracking_utils.py:96-100)."""
        out = []
        for g in range(0, len(tracklets), self.B):
            out.extend(self._run_group(tracklets[g:g + self.B]))
        return out


def run_overlapped(runners, tracklets):
    """Throughput form of the tracking loop: the tracklets are dealt to `len(runners)` TrackletRunners (each with its own
    buffers, model graph and HIP stream) whose lockstep groups advance ALTERNATELY — while the host waits for group A's
    boxes and computes its next crop bounds, group B's frame is on the device. Same per-tracklet results as
    runner.run(). Measured on one MI355X (round 3, scripts/probes/tracklet_groups_probe.py): 96 tracklets as 2 x 48 alternating
    10.7k frames/s against 9.8-10.3k one group after the other (144 / 192 tracklets as 3 / 4 groups: 10.7k / 10.5k) — the
    model graph of a 48-wide group fills the chip, so what overlaps is the host bubble (~0.5 ms per step) and the
    latency-bound sampling; splitting 48 tracklets into 2 x 24 loses more in half-filled graphs than it hides (round 2:
    7.8k against 8.4k). bench.py reports the single 48-wide group.
    tracklets: list of (clouds, boxes); returns the results in input order."""
    R = len(runners)
    dev = runners[0].device
    for r in runners:
        if r.stream is None:
            r.stream = torch.cuda.Stream(device=dev)
    # deal whole groups round-robin: runner k takes groups k, k+R, ...
    order, per_runner = [], [[] for _ in runners]
    pos = 0
    k = 0
    while pos < len(tracklets):
        b = runners[k % R].B
        per_runner[k % R].append((pos, tracklets[pos:pos + b]))
        pos += b
        k += 1
    results = [None] * len(tracklets)
    queues = [list(g) for g in per_runner]
    active = [None] * R                                             # (generator, start index) of the group in flight
    while any(queues) or any(a is not None for a in active):
        for r, runner in enumerate(runners):
            if active[r] is None and queues[r]:
                start, group = queues[r].pop(0)
                with torch.cuda.stream(runner.stream):
                    gen = runner._steps(group)
                    try:
                        next(gen)                                   # enqueue the first frame
                        active[r] = (gen, start)
                    except StopIteration as stop:                   # single-frame tracklets: nothing to track
                        for j, res in enumerate(stop.value):
                            results[start + j] = res
            elif active[r] is not None:
                gen, start = active[r]
                with torch.cuda.stream(runner.stream):
                    try:
                        next(gen)                                   # wait for this group's frame, enqueue its next one
                    except StopIteration as stop:
                        for j, res in enumerate(stop.value):
                            results[start + j] = res
                        active[r] = None
    return results
