"""Multi-process test plumbing: bounded spawn and a collectives object for ranks that share one card.

RCCL needs one GPU per rank, so tests that put two or three ranks on the single card of the test box run over
gloo.  gloo's own handling of device tensors (a pinned staging copy on a side stream that its worker thread
waits for) did not complete reliably with several processes on one card: buckets of the data-parallel trainer
and, once in a few runs, the small all-reduce inside the row-sharded frame stalled until the group timeout.
`HostStagedCollectives` therefore never hands gloo a device tensor: it waits for the device first -- so a
kernel that never finishes shows up as a stack inside `torch.cuda.synchronize`, not inside a collective --,
moves the buffer to the host, runs the collective there and copies the result back.  Every call appends a
line to a per-rank progress file when RST_TEST_COMM_LOG names a directory, which `spawn_bounded` prints when the
workers miss their deadline."""
import glob
import os
import socket
import time

import torch
import torch.distributed as dist

from superresolution_amd.parallel import Collectives


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Done:
    def wait(self):
        return True


class HostStagedCollectives(Collectives):
    def __init__(self, group=None, log_dir=None):
        super().__init__(group)
        self.n = 0
        log_dir = log_dir or os.environ.get("RST_TEST_COMM_LOG")
        self.log = open(os.path.join(log_dir, f"comm_rank{dist.get_rank()}.log"), "a", buffering=1) if log_dir else None

    def _note(self, what):
        self.n += 1
        if self.log:
            self.log.write(f"{time.time():.3f} #{self.n} {what}\n")

    @staticmethod
    def _host(t):
        if t.is_cuda:
            torch.cuda.synchronize(t.device)
            return t.cpu()
        return t

    def all_reduce(self, t, op=dist.ReduceOp.SUM, async_op=False):
        self._note(f"all_reduce {tuple(t.shape)} begin")
        h = self._host(t)
        dist.all_reduce(h, op=op, group=self.group)
        if h is not t:
            t.copy_(h)
        self._note("all_reduce end")
        return _Done() if async_op else None

    def broadcast(self, t, src=0):
        self._note(f"broadcast {tuple(t.shape)} begin")
        h = self._host(t)
        dist.broadcast(h, src=src, group=self.group)
        if h is not t:
            t.copy_(h)
        self._note("broadcast end")

    def exchange(self, pairs):
        self._note(f"exchange x{len(pairs)} begin")
        staged, land = [], []
        for send, recv, peer in pairs:
            s = None if send is None else self._host(send)
            r = None
            if recv is not None:
                r = torch.empty(recv.shape, dtype=recv.dtype) if recv.is_cuda else recv
                if r is not recv:
                    land.append((recv, r))
            staged.append((s, r, peer))
        super().exchange(staged)
        for dst, src in land:
            dst.copy_(src)
        self._note("exchange end")


def spawn_bounded(fn, args, nprocs, deadline_s=150.0, log_dir=None):
    """torch.multiprocessing.spawn that cannot hang: the workers are joined with a deadline and killed when it
    passes (a stuck rendezvous or collective then fails this one test instead of stalling the whole run).  The
    tails of the progress files under `log_dir` go into the failure message."""
    import torch.multiprocessing as mp
    if log_dir:
        os.environ["RST_TEST_COMM_LOG"] = str(log_dir)
    try:
        ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
        t_end = time.monotonic() + deadline_s
        try:
            while not ctx.join(timeout=5.0):
                if time.monotonic() > t_end:
                    for proc in ctx.processes:
                        if proc.is_alive():
                            proc.kill()
                    raise AssertionError(f"{nprocs} worker processes did not finish within {deadline_s:.0f} s; killed")
        except Exception as e:
            tails = []
            for f in sorted(glob.glob(os.path.join(str(log_dir), "comm_rank*.log"))) if log_dir else []:
                with open(f) as fh:
                    tails.append(os.path.basename(f) + ": " + " | ".join(l.strip() for l in fh.readlines()[-4:]))
            if tails:
                raise AssertionError(str(e) + "\nlast collectives per rank:\n" + "\n".join(tails)) from e
            raise
    finally:
        os.environ.pop("RST_TEST_COMM_LOG", None)
