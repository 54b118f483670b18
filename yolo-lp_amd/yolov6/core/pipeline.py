"""Several batches in flight on one GPU (new functionality, like ``sharded.py``; the reference runs one batch at a time).

At 32 images per GPU the deep layers of the detector cannot fill 256 CUs evenly (a 40x40 map is 400 workgroups for 512
resident slots, a 20x20 map 200), and a HIP stream runs its kernels one after the other.  ``InflightForward`` keeps
``depth`` engines -- each with its own activation arena and streams -- and gives consecutive batches to consecutive
engines, so that the workgroups of one batch's kernel fill the slots another batch's kernel leaves empty.  Every batch
still runs the complete forward; the kernels, their variants and therefore the results are those of a single engine.
Measured on yololps 640x640, 32 images per batch, fp16 (round 3, profiles/r03_inflight_lanes.txt): 13.0 k images/s with one
batch in flight, 16.0-16.3 k with four to twelve when every forward runs on one stream (``single_lane``, the default here);
with the three execution lanes per forward that one batch at a time prefers, 14.5-14.9 k.
"""
import torch

from yolov6.hip import runtime


class InflightForward:
    def __init__(self, model, depth=4, dtype=None, single_lane=None):
        p = next(model.parameters())
        if not p.is_cuda:
            raise RuntimeError('model is not on a GPU')
        self.device = p.device
        self.depth = max(1, int(depth))
        first = runtime.engine_for(model, dtype)
        with torch.no_grad():
            self.engines = [first] + [runtime.Engine.from_model(model, first.dtype, self.device) for _ in range(self.depth - 1)]
        with torch.cuda.device(self.device):
            self.streams = [torch.cuda.Stream(self.device) for _ in range(self.depth)]
        torch.cuda.synchronize(self.device)     # the engines' packed weights are uploaded before any side stream uses them
        # several forwards in flight: each on ONE stream (the fork / join events of the side lanes cost more than they hide
        # once other batches fill the gaps: six in flight 16.2 k images/s against 14.9 k, profiles/r03_inflight_lanes.txt)
        self.single_lane = True if single_lane is None else bool(single_lane)
        self._next = 0
        self._ws = [None] * self.depth          # detections-only path: one NMS workspace per engine ...
        self._ws_free = [None] * self.depth     # ... and the event after which its previous candidates are no longer needed

    def submit(self, x, fresh=True):
        """Enqueue the forward of batch ``x`` on the next engine; returns (pred, event): ``pred`` [B,N,290] fp32 is
        complete once ``event`` has fired (make the consumer's stream wait for it and call ``pred.record_stream``).
        ``fresh``: ``x`` was just produced on the caller's current stream, so the engine's stream must wait for that stream;
        pass False for inputs that have long been resident."""
        k = self._next
        self._next = (k + 1) % self.depth
        s = self.streams[k]
        shape = (x.shape[0], x.shape[2], x.shape[3])
        if k and shape not in self.engines[k].tuned and shape in self.engines[0].tuned:
            self.engines[k].copy_tuning(self.engines[0])      # tune once (engine 0), not once per engine
        self.engines[k].set_single_lane(self.single_lane)     # (engine 0 is shared with the model's one-at-a-time callers)
        if fresh:
            s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            pred = self.engines[k].forward(x)
            done = torch.cuda.Event()
            done.record(s)
        x.record_stream(s)
        return pred, done

    def submit_det(self, x, conf_thres, fresh=True):
        """Detections-only form of ``submit``: the forward of batch ``x`` writes NMS candidates (``Engine.forward_det``) into the
        slot's own workspace; returns (handle, event, release): run ``runtime.nms_candidates(handle, ...)`` on a stream that
        waits for ``event``, then call ``release(stream)`` so that the slot's next forward waits for that NMS."""
        k = self._next
        self._next = (k + 1) % self.depth
        s = self.streams[k]
        eng = self.engines[k]
        shape = (x.shape[0], x.shape[2], x.shape[3])
        if k and shape not in eng.tuned and shape in self.engines[0].tuned:
            eng.copy_tuning(self.engines[0])
        eng.set_single_lane(self.single_lane)
        if fresh:
            s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            if self._ws_free[k] is not None:
                s.wait_event(self._ws_free[k])          # the NMS of this slot's previous batch has read its candidates
                self._ws_free[k] = None
            if self._ws[k] is None or eng.bound != shape:
                # (a replaced workspace goes back to the caching allocator, which may hand the block out again on this stream at
                # once -- e.g. to the prediction tensor of the tuning forward below: only behind the wait above, because the
                # NMS that read it ran on the caller's stream)
                self._ws[k] = eng.det_workspace(*shape)
            handle = eng.forward_det(x, conf_thres, ws=self._ws[k])
            done = torch.cuda.Event()
            done.record(s)
        x.record_stream(s)

        def release(stream):
            ev = torch.cuda.Event()
            ev.record(stream)
            self._ws_free[k] = ev
        return handle, done, release
