"""hipGraph capture of a whole render step (forward + backward).

At the headline workload one step is ~10 kernel launches and ~0.25 ms of GPU time, while the Python / autograd side of
the same step costs ~0.3 ms on the host: eager execution is launch bound.  Capturing the step once and replaying the
graph removes the host from the loop (one hipGraphLaunch per step).  Every C-ABI entry point enqueues on the stream it
is given and neither allocates nor synchronises, so the library is capture-safe as is; torch's caching allocator
serves the capture from a private pool.

    step = GraphedStep(lambda: renderer(meshes, R, T).backward(cot), grads_of=(verts, maps, R, T))
    step()            # replays; verts.grad etc. are rewritten in place (static tensors)
"""
import torch


class GraphedStep:
    def __init__(self, fn, grads_of=(), warmup=3):
        self.fn, self.leaves = fn, tuple(grads_of)
        # the leaves' AccumulateGrad nodes were created on the default stream; running the backward on the capture
        # stream is intended here
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:
            quiet(False)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):  # warm up off the capture: lazy inits (module load, LDS attributes) happen here
            for _ in range(warmup):
                self._clear()
                fn()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self._clear()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other threads of the process (e.g. the RCCL watchdog of a multi-GPU run) may call into HIP
        # while this thread captures
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.out = fn()  # grads are created inside the capture: static tensors from the graph's pool

    def _clear(self):
        for t in self.leaves:
            t.grad = None

    def __call__(self):
        self.graph.replay()
        return self.out
