"""Data-parallel sharding for the hot path: one process per GPU, clouds sharded across ranks, one
gradient all-reduce per step over RCCL/xGMI (`torch.distributed` backend "nccl" on ROCm).

The reference does this with DistributedDataParallel + DistributedSampler and nothing else
(train_modelnet_dist.py:117-125,206: `broadcast_buffers=False`, no SyncBN), i.e. clouds are
independent units and the only exchange is the parameter-gradient mean.  Here:

  * `shard_range(n, rank, world)`  contiguous shard of n clouds/scenes for this rank;
  * `GradientSynchronizer`         flat fp32 buckets filled from autograd hooks as gradients become
                                   ready (reverse registration order ~ backward order); each full
                                   bucket is all-reduced asynchronously while backward continues, so
                                   the long early-stage backward hides the transfer.  Bucket size is
                                   chosen for xGMI: per-link bandwidth ~153 GB/s means a 32 MiB bucket
                                   is ~0.4 ms on a ring -- large enough to amortise launch latency,
                                   small enough that the last bucket's tail is short;
  * `allreduce_gradients`          one-shot flat all-reduce for small parameter sets.
BatchNorm statistics stay per-rank, as in the reference.
"""
import os
import socket
import sys

import torch
import torch.distributed as dist


def torchrun_command(script, argv, nproc, port=None, python=None):
    """The command line that runs `script argv...` as `nproc` ranks of ONE node, one process per GPU -- the launch the
    reference documents for its trainers (`python -m torch.distributed.launch --nproc_per_node N`,
    pytorch/README.md:70-76, function/train_modelnet_dist.py:117-125) in its present-day spelling, and exactly what
    scripts/scale.sh and the round driver type by hand.  127.0.0.1 on purpose: a container's hostname may not resolve."""
    if port is None:
        with socket.socket() as s:  # a free port now; torchrun binds it a moment later (single node: good enough)
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(nproc)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), script] + list(argv)


def self_launch(script, argv, nproc, visible_devices=None):
    """`python bench.py --gpus N` typed into a bare shell (no WORLD_SIZE in the environment): replace this process by
    the N-rank launch of the same command.  With fewer than N devices visible (a 1-GPU box) every rank is put on
    device 0 over gloo -- CL3D_BENCH_ONE_DEVICE=1, a stand-in that exercises the N > 1 code path and says so in its
    JSON line (`backend`, `device`, `one_device_standin`); it is not a scaling measurement.  Never returns."""
    if visible_devices is None:
        visible_devices = torch.cuda.device_count()
    env = dict(os.environ)
    if visible_devices < nproc:
        env["CL3D_BENCH_ONE_DEVICE"] = "1"
        print(f"{os.path.basename(script)}: {nproc} ranks asked for, {visible_devices} device(s) visible -- all ranks on "
              f"device 0 over gloo (stand-in for the code path, not a scaling run)", file=sys.stderr, flush=True)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    cmd = torchrun_command(script, argv, nproc)
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def prepare_environment(environ=None):
    """What every rank's environment needs BEFORE the HIP runtime starts (before the first torch.cuda call), whoever
    launched it: HSA_ENABLE_IPC_MODE_LEGACY=0 -- this driver only supports dmabuf IPC, and without it RCCL's
    cross-process buffer registration fails with `hipIpcGetMemHandle: invalid argument`.  self_launch() sets it for the
    ranks it starts; a `torchrun ... bench.py --gpus 8` typed by someone else reaches the benches through this call.
    An explicit setting in the environment is left alone."""
    env = os.environ if environ is None else environ
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def device_identity(device):
    """A string that two ranks share exactly when they sit on the SAME physical GPU: the device's UUID where the
    runtime reports one, else its PCI bus id, else (nothing better) its ordinal and name."""
    props = torch.cuda.get_device_properties(device)
    uuid = getattr(props, "uuid", None)
    if uuid is not None and str(uuid).strip("0-") != "":
        return f"uuid:{uuid}"
    bus = [getattr(props, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
    if all(b is not None for b in bus):
        return "pci:%04x:%02x:%02x" % tuple(int(b) for b in bus)
    return f"ordinal:{torch.cuda.current_device() if device is None else torch.device(device).index}:{props.name}"


def rank_census(identity, group=None):
    """[{rank, device}] of every rank of the job (all_gather_object), and the proof the JSON line carries that N ranks
    sat on N DISTINCT devices: under the RCCL backend two ranks on one device raise here -- such a run is not a
    scaling measurement and RCCL may deadlock on it; under gloo (the one-device stand-in) the list says so."""
    world = dist.get_world_size(group)
    got = [None] * world
    dist.all_gather_object(got, {"rank": dist.get_rank(group), "device": identity}, group=group)
    got.sort(key=lambda r: r["rank"])
    devices = [r["device"] for r in got]
    if dist.get_backend(group) == "nccl" and len(set(devices)) != world:
        shared = sorted({d for d in devices if devices.count(d) > 1})
        raise RuntimeError(f"data-parallel run over RCCL with ranks sharing a device: {shared} -- one process per GPU "
                           f"(LOCAL_RANK must select distinct devices); census: {got}")
    return got


def shard_range(n, rank, world):
    """[lo, hi) of rank's contiguous shard of n items; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _mean_inplace(flat, world, group, async_op=False):
    backend = dist.get_backend(group)
    if backend == "nccl":
        return dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group, async_op=async_op), False
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op), True


def allreduce_gradients(params, world, group=None):
    """Average .grad of `params` across ranks with a single flat all-reduce.

    The flat buffer covers EVERY parameter that requires a gradient, with zeros where this rank produced none
    (a head that saw no sample of its shape), so its length is the same on all ranks; afterwards every such
    parameter has a .grad (the mean), as DistributedDataParallel leaves it -- otherwise only some replicas
    would step that parameter and they would drift apart silently."""
    params = [p for p in params if p.requires_grad]
    if not params or world == 1:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in params])
    _, need_div = _mean_inplace(flat, world, group)
    if need_div:
        flat.div_(world)
    off = 0
    for p in params:
        n = p.numel()
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        p.grad.copy_(flat[off:off + n].view_as(p))
        off += n


class FlatGradients:
    """All gradients of `params` as views of ONE flat fp32 buffer (assigned as `.grad`), so the exchange is a
    single in-place all-reduce with no packing and the buffer can be zeroed inside a captured step.

        flat = FlatGradients(params)
        # step:  flat.zero_()  ...forward/backward accumulate in place...  flat.allreduce_mean(world)
    """

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else None
        self.buffer = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.buffer[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_(self):
        self.buffer.zero_()

    def allreduce_mean(self, world, group=None):
        if world == 1 or self.buffer.numel() == 0:
            return
        _, need_div = _mean_inplace(self.buffer, world, group)
        if need_div:
            self.buffer.div_(world)


class GradientSynchronizer:
    """Bucketed, backward-overlapped gradient averaging.

        sync = GradientSynchronizer(model.parameters(), world)
        loss.backward()      # hooks launch async all-reduces bucket by bucket
        sync.finish()        # wait, scale, scatter back into .grad

    Collectives must be issued in the same order on every rank.  Which parameters receive a gradient may differ
    between ranks (per-shape heads), so "bucket complete" is a rank-local event; the launch order is therefore
    FIXED: bucket b is launched only once buckets 0..b-1 have been, from a hook when that happens during the
    backward pass (the usual case: buckets fill in backward order) and otherwise from finish(), which launches
    whatever is left in index order with zeros standing in for absent gradients.  finish() gives every parameter
    a .grad (the mean over ranks), as DistributedDataParallel does.
    """

    def __init__(self, params, world, bucket_bytes=32 << 20, group=None):
        self.world, self.group = world, group
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []  # each: dict(params, offsets, flat, filled, pending, handle)
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur)
        self._slot_of = {}  # id(param) -> (bucket index, index within bucket)
        for bi, b in enumerate(self.buckets):
            for i, p in enumerate(b["params"]):
                self._slot_of[id(p)] = (bi, i)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params] if world > 1 else []
        self._need_div = False
        self._next = 0  # index of the next bucket to launch (launch order == bucket order on every rank)

    def _close(self, plist):
        total = sum(p.numel() for p in plist)
        offs, o = [], 0
        for p in plist:
            offs.append(o)
            o += p.numel()
        self.buckets.append(dict(params=list(plist), offsets=offs,
                                 flat=torch.zeros(total, dtype=torch.float32, device=plist[0].device),
                                 filled=[False] * len(plist), pending=len(plist), handle=None))

    def _launch_ready(self, force=False):
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if b["pending"] != 0:
                if not force:
                    return
                for i, p in enumerate(b["params"]):  # no gradient on this rank this step: contribute zeros
                    if not b["filled"][i]:
                        b["flat"][b["offsets"][i]:b["offsets"][i] + p.numel()].zero_()
            b["handle"], self._need_div = _mean_inplace(b["flat"], self.world, self.group, async_op=True)
            self._next += 1

    def _on_grad(self, p):
        bi, i = self._slot_of[id(p)]
        b = self.buckets[bi]
        if b["handle"] is not None or b["filled"][i]:
            raise RuntimeError("GradientSynchronizer: a gradient arrived for a bucket that is already in flight; "
                               "call finish() once per backward pass")
        n = p.numel()
        b["flat"][b["offsets"][i]:b["offsets"][i] + n].copy_(p.grad.reshape(-1))
        b["filled"][i] = True
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch_ready()

    def finish(self):
        if self.world == 1:
            return
        self._launch_ready(force=True)
        for b in self.buckets:
            b["handle"].wait()
            if self._need_div:
                b["flat"].div_(self.world)
            for i, p in enumerate(b["params"]):
                n = p.numel()
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                p.grad.copy_(b["flat"][b["offsets"][i]:b["offsets"][i] + n].view_as(p.grad))
            b["pending"], b["handle"], b["filled"] = len(b["params"]), None, [False] * len(b["params"])
        self._next = 0

    def remove(self):
        for h in self._hooks:
            h.remove()
