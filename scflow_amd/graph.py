"""hipGraph capture of the whole refinement pass.

At batch 1 (BASELINE configs[1]) one ``get_pose`` is ~430 kernel launches of a few
microseconds each: eager execution is bound by host-side launch overhead, not by the GPU.
Every C-ABI entry point only enqueues work on the stream it is given and never synchronises or
allocates, so the entire pass -- three encoder passes, correlation build, 8 x (lookup, motion
encoder, GRU, heads, pose head, pose update, re-projection) -- is captured once into a HIP
graph (``torch.cuda.CUDAGraph`` = hipGraph on ROCm) and replayed with a single launch.

Shapes are static per instance; inputs are copied into persistent device buffers before each
replay and the outputs are persistent too (clone them if they must outlive the next call).
A caller that produces its inputs on the device writes them straight into ``static_in[name]`` (the
buffers the graph reads) and calls the instance without arguments: no copies, one graph launch.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

__all__ = ['GraphedRefiner']

_INPUTS = ('render_images', 'real_images', 'ref_rotation', 'ref_translation', 'depth',
           'internel_k', 'label')


class GraphedRefiner:
    def __init__(self, model, example: Dict[str, torch.Tensor], warmup: int = 2) -> None:
        if warmup < 1:
            # the eager pass packs the weights, creates the cached constants (ops.constant refuses to be first called
            # inside a capture), the side streams and the per-shape dispatch plans: none of that may happen in the graph
            raise ValueError('GraphedRefiner needs at least one eager warm-up pass (warmup >= 1)')
        self.model = model
        self.static_in = {k: example[k].clone().contiguous() for k in _INPUTS}
        # warm-up and capture run on ONE private stream: the side stream that belongs to it
        # (ops.side_stream keys side streams by main stream) is created during the warm-up, never
        # inside the capture
        cs = torch.cuda.Stream()
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):              # warm-up: packs weights, primes the allocator
            for _ in range(warmup):
                self._run()
        torch.cuda.current_stream().wait_stream(cs)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=cs):
            self.static_out = self._run()
        torch.cuda.synchronize()

    def _run(self):
        s = self.static_in
        return self.model.get_pose(s['render_images'], s['real_images'], s['ref_rotation'],
                                   s['ref_translation'], s['depth'], s['internel_k'], s['label'])

    def __call__(self, inputs: Optional[Dict[str, torch.Tensor]] = None):
        """replay on (optionally new) inputs -> the reference's 7-tuple of per-iteration lists
        (persistent buffers, overwritten by the next call)."""
        if inputs is not None:
            # one multi-tensor copy per dtype (six fp32 inputs + the int64 labels: two launches instead of seven)
            todo = [k for k in _INPUTS if inputs[k] is not self.static_in[k]]
            groups = {}
            for k in todo:
                src, dst = inputs[k], self.static_in[k]
                if src.dtype == dst.dtype and src.device == dst.device and src.shape == dst.shape:
                    groups.setdefault(dst.dtype, ([], []))
                    groups[dst.dtype][0].append(dst)
                    groups[dst.dtype][1].append(src.contiguous())
                else:
                    dst.copy_(src, non_blocking=True)
            for dsts, srcs in groups.values():
                torch._foreach_copy_(dsts, srcs)
        self.graph.replay()
        return self.static_out
