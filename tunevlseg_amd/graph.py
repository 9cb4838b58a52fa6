"""Forward + backward of a train step as a hipGraph (``torch.cuda.graph``): one graph launch instead of 340 (VPT) - 820 (CRIS) kernel launches.

Opt-in (``bench.py --graph``, ``Trainer(graph_step=True)``): the step is GPU-bound on one device, the graph frees the host
(CRIS: 33 -> 12 ms of Python per step) for the input pipeline and for eight ranks sharing one node's cores.

What makes the capture legal (the two failed attempts of rounds 2 and 3 are in DESIGN.md §7):

* every launch of the library goes to torch's current stream (``hip._stream``), side streams fork from and join it with events
  (``nets.towers.SideStream``) -- all of it capturable as it stands;
* the AccumulateGrad nodes of the flat parameter views are created when the optimiser registers its gradient-exchange hooks, and autograd
  runs each node on the stream that was current THEN.  Built on the default stream, they drag the default stream into the capture, nothing
  joins it back and ``hipStreamEndCapture`` dies.  Hence :func:`use_private_stream` -- call it before building the module and the optimiser;
  :class:`GraphedStep` refuses to capture on the default stream;
* the optimiser update stays outside the graph (its bias correction takes the step count as a launch argument), so does the gradient
  exchange of a multi-GPU job (``opt.step()`` launches the buckets the disarmed hooks did not);
* the slot pool of the tagged ``atomicMax`` hand-over (``hip._max_slot``) is cleared inside the captured region: a replay re-uses the slots
  and tags of the capture;
* nothing in the step may take a per-step launch argument from the host.  The one such thing in the package is train-mode dropout
  (``ops.DropoutFn``: the SharedAttn learner's TransformerEncoderLayer): a step that used it is not kept as a graph and its shape stays
  eager (``GraphedStep._eager_only``);
* batches may carry host entries (the reference's collate leaves ``mask_shape`` on the CPU): only device tensors are keyed and copied.
"""
from __future__ import annotations

from typing import Any, Mapping

import torch

from . import hip, ops


def use_private_stream(device=None) -> "torch.cuda.Stream":
    """Make a fresh non-default stream the current one for this thread (idempotent per call site: call once, before the module exists)."""
    cur = torch.cuda.current_stream(device)
    if cur != torch.cuda.default_stream(device):
        return cur
    s = torch.cuda.Stream(device)
    s.wait_stream(cur)
    torch.cuda.set_stream(s)
    return s


def _device_tensors(batch: Mapping[str, Any]) -> dict:
    """The entries a captured step can read: device tensors.  What the data pipeline leaves on the host (``mask_shape`` of the reference's
    collate, file names) is only read by Python outside the step, so it is neither part of the shape key nor copied per replay."""
    tensors = {k: v for k, v in batch.items() if isinstance(v, torch.Tensor)}
    on_device = {k: v for k, v in tensors.items() if v.is_cuda}
    return on_device or tensors   # (a batch without any device tensor -- the host-side unit tests of the key -- keeps all of them)


def _key(batch: Mapping[str, Any]):
    return tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(_device_tensors(batch).items()))


class GraphedStep:
    """``loss = stepper(batch)`` == ``opt.zero_grad(); loss = module.training_step(batch, 0); loss.backward()``; the caller then runs
    ``opt.step()``.  The first batch of a shape runs eagerly (it fills the caches: position tables, constant index tensors, workspaces), the
    second is captured, later ones copy their tensors into the captured step's inputs and replay.  Ragged text batches give several shapes:
    at most ``max_graphs`` are kept (each holds a private pool with a step's activations); other shapes stay eager.
    The returned loss of a replay is the captured step's output tensor: it is overwritten by the next replay (add it to a running sum, as
    ``Trainer.fit`` does, before calling again)."""

    def __init__(self, module, opt, max_graphs: int = 2):
        self.module, self.opt, self.max_graphs = module, opt, max_graphs
        self._seen: set = set()
        self._graphs: dict = {}
        self._eager_only: set = set()   # shapes whose step takes per-step launch arguments from the host (train-mode dropout seeds)
        self.replays = 0

    def _eager(self, batch):
        self.opt.zero_grad()   # (an eager step keeps its backward-overlapped gradient exchange: the hooks are armed outside a capture)
        loss = self.module.training_step(batch, 0)
        loss.backward()
        return loss

    def __call__(self, batch: Mapping[str, Any]):
        key = _key(batch)
        entry = self._graphs.get(key)
        if entry is None:
            if key not in self._seen or key in self._eager_only or len(self._graphs) >= self.max_graphs:
                self._seen.add(key)
                return self._eager(batch)
            entry = self._capture(batch)
            if entry is None:   # the step draws a fresh dropout mask per call: a replay would repeat the captured one
                self._eager_only.add(key)
                return self._eager(batch)
            self._graphs[key] = entry
        g, static, loss, appended = entry
        for k, v in static.items():
            v.copy_(batch[k], non_blocking=True)
        g.replay()
        for metric, attr, tensors in appended:   # what the step's Python appended to the metric states (the batch's confusion counts): a copy per replay
            getattr(metric, attr).extend(t.clone() for t in tensors)   # (looked up now: reset() installs a new list every epoch)
        self.replays += 1
        return loss

    def _capture(self, batch):
        on_device = _device_tensors(batch)
        if not any(v.is_cuda for v in on_device.values()):
            raise RuntimeError("GraphedStep: the batch holds no device tensor")
        dev = next(iter(on_device.values())).device
        stream = torch.cuda.current_stream(dev)
        if stream == torch.cuda.default_stream(dev):
            raise RuntimeError("GraphedStep: the current stream is the default stream -- call tunevlseg_amd.graph.use_private_stream() before building "
                               "the module and its optimiser (the AccumulateGrad nodes must live on the capture stream)")
        static = {k: v.clone() for k, v in on_device.items()}
        step_in = {**batch, **static}
        # metric states are python lists of per-batch count tensors (task.DiceSamples / JaccardBinary): note what the captured step appends
        lists = [(m, a) for m in getattr(self.module, "metrics", {}).values() for a, lst in vars(m).items() if isinstance(lst, list)]
        before = [len(getattr(m, a)) for m, a in lists]
        arm = getattr(self.opt, "set_exchange_armed", lambda armed: None)
        arm(False)   # no collective inside the graph: opt.step() -> exchange.finish() launches every bucket after the replay
        dropout_calls = ops._dropout_calls
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                for pool in hip._MAX_SLOTS.values():
                    pool[0].zero_()
                loss = self._eager(step_in)
        finally:
            arm(True)
        if ops._dropout_calls != dropout_calls:
            # ops.DropoutFn passes its (seed, call index) as a launch argument: frozen into the graph, every replay would re-use the capture's
            # mask and the training dynamics would silently differ from the eager step.  Drop the graph; this shape stays eager, with the
            # counter put back so that the eager steps draw the masks they would have drawn without the attempt.
            ops._dropout_calls = dropout_calls
            for (m, a), n0 in zip(lists, before):
                del getattr(m, a)[n0:]
            del g
            return None
        appended = []
        for (m, a), n0 in zip(lists, before):   # the capture did not run: its appends come back per replay, as copies of the captured step's outputs
            lst = getattr(m, a)
            if len(lst) > n0:
                appended.append((m, a, lst[n0:]))
                del lst[n0:]
        return g, static, loss, appended
