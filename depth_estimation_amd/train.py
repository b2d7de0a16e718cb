"""Per-GPU data-parallel training loop around the fused photometric stack.

This is the build's counterpart of the reference's trainer step
(/root/reference/train.py:96-153: forward, loss, zero_grad / backward / step),
re-shaped for one process per GPU: every rank owns a shard of the batch, runs
depth + pose networks (stock PyTorch convolutions -- out of the hot-path scope)
and the HIP view-synthesis loss on it, and the only exchange is the all-reduce of a
flat float32 gradient bucket plus the scalar loss (RCCL over xGMI on a GPU
node; gloo on CPU for tests), issued in a few chunks from gradient hooks so
that it overlaps the rest of backward.  The reference itself is single-process
(train.py:348) and hard-wired to batch 1.

    python -m torch.distributed.run --nproc-per-node N -m depth_estimation_amd.train --steps 20

The loss has no CPU implementation: on a machine without a GPU the trainer
must be given a `loss_fn` (the tests pass the oracle), otherwise it raises.
"""
from __future__ import annotations

import argparse
import os
import time
from typing import Callable, Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from .synth import Sequence, make_sequence, pose_to_matrix

LossFn = Callable[..., torch.Tensor]


def conv(cin, cout, stride=1):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, stride, 1), nn.ELU(inplace=True))


class DepthNet(nn.Module):
    """Small encoder-decoder; sigmoid disparities at 4 scales (full, 1/2, 1/4, 1/8)."""

    def __init__(self, width: int = 16):
        super().__init__()
        w = width
        self.enc = nn.ModuleList([conv(3, w, 1), conv(w, 2 * w, 2), conv(2 * w, 4 * w, 2),
                                  conv(4 * w, 8 * w, 2)])
        self.dec = nn.ModuleList([conv(8 * w, 4 * w), conv(4 * w, 2 * w), conv(2 * w, w)])
        self.heads = nn.ModuleList([nn.Conv2d(c, 1, 3, 1, 1) for c in (w, 2 * w, 4 * w, 8 * w)])

    def forward(self, x) -> List[torch.Tensor]:
        feats = []
        for layer in self.enc:
            x = layer(x)
            feats.append(x)
        disp = [None] * 4
        y = feats[3]
        disp[3] = torch.sigmoid(self.heads[3](y))
        for k, s in enumerate((2, 1, 0)):
            y = F.interpolate(self.dec[k](y), scale_factor=2, mode="nearest") + feats[s]
            disp[s] = torch.sigmoid(self.heads[s](y))
        return disp


class PoseNet(nn.Module):
    """(target, source) pair -> 6-DoF motion, scaled small as usual."""

    def __init__(self, width: int = 16):
        super().__init__()
        w = width
        self.net = nn.Sequential(conv(6, w, 2), conv(w, 2 * w, 2), conv(2 * w, 4 * w, 2),
                                 conv(4 * w, 4 * w, 2), nn.Conv2d(4 * w, 6, 1))

    def forward(self, a, b):
        out = self.net(torch.cat([a, b], 1)).mean((2, 3)) * 0.01
        return out[:, :3], out[:, 3:]


class FlatGradBucket:
    """All parameter gradients as views into one contiguous float32 buffer (slot 0 holds the
    scalar loss), so the data-parallel exchange needs no packing copies.  The buffer is cut into
    `n_chunks` contiguous ranges in parameter order.  Backward fills them roughly last-to-first;
    a hook on every parameter counts its chunk down and, when the chunk's last gradient has been
    accumulated, starts that chunk's all-reduce at once (asynchronously, RCCL's own stream), so
    the exchange of the late layers runs while backward is still working on the early ones.
    `finish()` waits for the chunks and turns the sums into means.  The hooks act only between
    `begin()` and `finish()`; a backward pass outside that bracket leaves the exchange to the
    caller (`all_reduce_mean`: the whole buffer at once, blocking)."""

    def __init__(self, params, n_chunks: int = 4):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(1 + n, dtype=torch.float32, device=dev)  # slot 0: the loss
        o = 1
        bounds = []
        for p in self.params:
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            bounds.append((o, o + p.numel()))
            o += p.numel()
        # chunk c = parameters [first[c], first[c + 1]); chunk 0 also carries the loss slot
        n_chunks = max(1, min(n_chunks, len(self.params)))
        target, first, acc = n / n_chunks, [0], 0
        for i, p in enumerate(self.params[:-1]):
            acc += p.numel()
            if acc >= target * len(first) and len(first) < n_chunks:
                first.append(i + 1)
        self._chunk_of = []
        self._ranges = []
        for c, lo in enumerate(first):
            hi = first[c + 1] if c + 1 < len(first) else len(self.params)
            self._chunk_of += [c] * (hi - lo)
            self._ranges.append((0 if c == 0 else bounds[lo][0], bounds[hi - 1][1]))
        self._size = [self._chunk_of.count(c) for c in range(len(first))]
        self._left = list(self._size)
        self._handles = []
        self._armed = False  # set by begin(), cleared by finish(): only then do the hooks start exchanges
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(lambda _p, c=self._chunk_of[i]: self._grad_ready(c))

    @staticmethod
    def _world() -> int:
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def begin(self, loss: torch.Tensor):
        """Call before backward: clears the counters and stores the loss in its slot."""
        self._left = list(self._size)
        self._handles = []
        self._armed = True
        self.flat[0] = loss.detach()

    @staticmethod
    def _exchanging() -> bool:
        """True inside a process group, also one of a single rank: the collectives then run as they
        do on a node (a one-rank all-reduce leaves the data as it is)."""
        return dist.is_available() and dist.is_initialized()

    def _grad_ready(self, c: int):
        if not self._armed:
            return
        self._left[c] -= 1
        if self._left[c] == 0 and self._exchanging():
            lo, hi = self._ranges[c]
            self._handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def zero(self):
        self.flat.zero_()

    def finish(self) -> torch.Tensor:
        """Call after backward: waits for the chunk all-reduces, averages; returns the mean loss."""
        world = self._world()
        if not self._armed:
            raise RuntimeError("FlatGradBucket.finish() without begin(): no exchange was started "
                               "(use all_reduce_mean for a blocking exchange after a plain backward)")
        self._armed = False
        if self._exchanging():
            if any(n != 0 for n in self._left):  # a parameter got no gradient this step: its chunk never fired
                for h in self._handles:
                    h.wait()
                self._handles = [dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True)
                                 for c, (lo, hi) in enumerate(self._ranges) if self._left[c] != 0]
            for h in self._handles:
                h.wait()
            self._handles = []
            self.flat.mul_(1.0 / world)
        return self.flat[0]

    def all_reduce_mean(self, loss: torch.Tensor) -> torch.Tensor:
        """One blocking exchange of the whole buffer after a backward pass that was not bracketed by
        begin() / finish() (no overlap with backward).  Refused between begin() and finish(): the
        hooks have then started chunk exchanges of their own, and a second sum over the same
        memory would count those chunks twice."""
        if self._armed:
            raise RuntimeError("FlatGradBucket.all_reduce_mean() between begin() and finish(): "
                               "the chunk exchanges are already running; call finish()")
        self.flat[0] = loss.detach()
        if self._exchanging():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.mul_(1.0 / self._world())
        return self.flat[0]


class Trainer:
    def __init__(self, device, loss_fn: Optional[LossFn] = None, lr: float = 1e-4, width: int = 16,
                 seed: int = 0, amp: bool = False):
        self.device = torch.device(device)
        # amp: the networks run under bfloat16 autocast (their convolutions are stock PyTorch and out of
        # this build's scope); the view-synthesis loss always gets and returns float32
        self.amp = amp
        if loss_fn is None:
            if self.device.type != "cuda":
                from ._lib import DepthwarpError
                raise DepthwarpError("the view-synthesis loss runs on ROCm GPUs only; pass loss_fn "
                                     "explicitly to drive the loop on another device")
            from .ops import PlanLoss
            loss_fn = PlanLoss()  # pre-allocated workspace and gradient buffers, reused every step
        self.loss_fn = loss_fn
        torch.manual_seed(seed)  # identical initial weights on every rank
        self.depth_net = DepthNet(width).to(self.device)
        self.pose_net = PoseNet(width).to(self.device)
        params = list(self.depth_net.parameters()) + list(self.pose_net.parameters())
        self.bucket = FlatGradBucket(params)
        self.opt = torch.optim.Adam(self.bucket.params, lr=lr)

    def forward_loss(self, seq: Sequence) -> torch.Tensor:
        tgt, prev, nxt = seq.color
        with torch.autocast(self.device.type, dtype=torch.bfloat16, enabled=self.amp):
            disp = self.depth_net(tgt)
            poses = [self.pose_net(prev, tgt), self.pose_net(tgt, nxt)]
        disp = [d.float().contiguous() for d in disp]
        Ts = []
        for f, (aa, tr) in enumerate(poses):
            # frame -1 is predicted in temporal order and inverted, as is customary
            Ts.append(pose_to_matrix(aa.float(), tr.float(), invert=(f == 0)))
        T = torch.stack(Ts, 1).contiguous()
        return self.loss_fn(disp, seq.color, seq.color_pyr, seq.K, seq.inv_K, T, None)

    def step(self, seq: Sequence) -> torch.Tensor:
        """One optimisation step on this rank's shard; returns the loss averaged over the ranks
        (a 0-dim device tensor: reading it is the caller's synchronisation, not the step's)."""
        self.bucket.zero()
        loss = self.forward_loss(seq)
        self.bucket.begin(loss)
        loss.backward()                    # chunk all-reduces start from the gradient hooks
        mean_loss = self.bucket.finish()
        self.opt.step()
        return mean_loss


def shard(seq: Sequence, rank: int, world: int) -> Sequence:
    B = seq.color[0].shape[0]
    assert B % world == 0, "global batch must divide evenly over the ranks"
    n = B // world
    sl = lambda t: None if t is None else t[rank * n:(rank + 1) * n].contiguous()
    return Sequence([sl(d) for d in seq.disp], [sl(c) for c in seq.color],
                    [sl(c) for c in seq.color_pyr], sl(seq.K), sl(seq.inv_K), sl(seq.T),
                    sl(seq.tie_noise))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=12, help="sequences per GPU")
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    args = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if "WORLD_SIZE" in os.environ:
        dist.init_process_group("nccl", device_id=dev)
    trainer = Trainer(dev)
    seq = make_sequence(args.batch, args.height, args.width, seed=rank).to(dev)
    trainer.step(seq)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(seq)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    if rank == 0:
        print(f"loss {float(loss):.5f}  {1e3 * dt:.2f} ms/step  "
              f"{3 * args.batch * world / dt:.0f} frames/s on {world} GPU(s)")
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
