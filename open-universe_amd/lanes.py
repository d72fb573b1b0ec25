"""
K `enhance` calls in flight side by side on ONE device, inside one process.

Why: the reference's real workload is a directory of files of arbitrary lengths walked one by one
(open_universe/bin/enhance.py:173-192).  Files of different lengths cannot share a batch without changing their result
(the reference has no mask: `max_collator`, datasets/datamodule.py:24-42), and ONE batch-1 enhance cannot fill an MI355X:
a third of it is the GRU recurrence on 64 of 256 CUs, the rest are launches of a few hundred tiles each.  K independent
calls on K streams fill those gaps with each other's kernels -- every utterance is still computed by exactly the kernels,
tilings and summation orders of the one-at-a-time call, so the results are bit-identical to the serial loop's.

How: lane k = a model object of its own on the SAME packed weights (`Universe.fork()`: own C handle -- a handle is not
re-entrant --, own workspace, own status record) + a stream of its own.  Calls are enqueued round-robin from ONE host
thread (an enhance call never synchronises in free-running mode; the host needs ~1.3 ms to enqueue what the device runs in
7 ms).  The library is told the lane layout (`ou_set_lanes`) because the workgroups of a GRU cluster wait for each other:
all GRU launches that can be on the device at a time have to fit there whole, and lane k gets XCDs of its own for its
clusters.
"""
import torch


class LanePool:
    """`lanes` models on one set of weights, one stream each.  Use as a context manager or call close()."""

    MAX_LANES = 8

    def __init__(self, model, lanes, max_batch=0):
        """`max_batch`: the largest batch size any call submitted to this pool will have, when the calls differ in size
        (groups of a ragged set, files of different channel counts) -- all lanes then agree on how the GRU clusters of the
        pool share the XCDs (include/ouniverse.h, ou_set_lane_batch).  0: all calls have one size."""
        lanes = int(lanes)
        if not 1 <= lanes <= self.MAX_LANES:
            raise ValueError(f"lanes must be in [1, {self.MAX_LANES}]")
        if max_batch and max_batch > 1:
            # every lane's GRU clusters have to be resident together: no more lanes than the device holds at this batch size
            lanes = max(1, min(lanes, int(model._L.ou_lane_capacity(model._handle, int(max_batch)))))
        self.device = model.device
        # the forks (a C handle + workspaces each) are kept on the primary model and re-used by later pools
        forks = model.__dict__.setdefault("_lane_forks", [])
        while len(forks) < lanes - 1:
            forks.append(model.fork())
        self.models = [model] + forks[:lanes - 1]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(lanes)]
        self._saved_mode = model.check_status
        for k, m in enumerate(self.models):
            m.set_lanes(lanes, k, max_batch)
            m.check_status = False  # free-running: nothing inside a lane waits for the device
        self._next = 0
        self._busy = [False] * lanes
        self._closed = False

    @property
    def lanes(self):
        return len(self.models)

    def submit(self, fn, *inputs):
        """Run `fn(model)` -- device work only, e.g. `lambda m: m.enhance(x, rng=g)` -- on the next lane's stream and return
        (lane index, whatever fn returned).  Nothing is waited for; the result may be used after `wait(lane)` /
        `synchronize()`.  `inputs`: device tensors fn reads that were produced on the caller's stream (the lane waits for
        that stream and the tensors are kept from being recycled until the lane is done with them)."""
        k = self._next
        self._next = (k + 1) % len(self.models)
        s = self.streams[k]
        s.wait_stream(torch.cuda.current_stream(self.device))
        for t in inputs:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(s)
        self._busy[k] = True  # (before fn: if it raises half-way, what it did enqueue is still waited for at close)
        with torch.cuda.stream(s):
            out = fn(self.models[k])
        return k, out

    def wait(self, lane):
        """Block until lane `lane` is idle; raises if one of its calls flagged a device-side time-out."""
        if not self._busy[lane]:
            return
        with torch.cuda.stream(self.streams[lane]):
            self.models[lane].synchronize()
        self._busy[lane] = False

    def synchronize(self):
        for k in range(len(self.models)):
            self.wait(k)

    def close(self):
        if self._closed:
            return
        self._closed = True
        try:
            self.synchronize()
        finally:
            for st in self.streams:  # whatever happened above: nothing of this pool may still run when the models go back
                st.synchronize()
            for m in self.models:
                m.set_lanes(1, 0)
                m.check_status = self._saved_mode
            for m in self.models[1:]:  # the forks stay (handles are cheap), their workspaces -- grown to the longest call -- go
                try:
                    m.reset_workspace()
                except RuntimeError:
                    pass  # (a time-out on that lane has been raised by synchronize() above already)
            self.models = self.models[:1]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
