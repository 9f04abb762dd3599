"""One process per GPU, independent proofs per rank, no data-path collective.  torch.distributed is
used for exactly two things: the barrier that brackets the timed region and the MAX over ranks of
the elapsed time (bench.py contract).  Backend: "nccl" (= RCCL) on GPUs, "gloo" on CPU (tests)."""
import os
import time


class Group:
    def __init__(self, rank=0, local_rank=0, world=1, dist=None, device=None):
        self.rank, self.local_rank, self.world, self.dist, self.device = rank, local_rank, world, dist, device

    def barrier(self):
        if self.dist is None:
            return
        if self.device is not None:            # nccl: name the device, or the barrier picks one by rank and may warn / stall
            self.dist.barrier(device_ids=[self.device.index])
        else:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if self.dist is None:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def init_from_env(use_cuda, backend=None, force_device=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run sets them."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return Group(rank, local_rank, world)
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    device = None
    backend = backend or ("nccl" if use_cuda else "gloo")
    if use_cuda:
        d = local_rank if force_device is None else force_device
        torch.cuda.set_device(d)
        device = torch.device("cuda", d) if backend == "nccl" else None
    dist.init_process_group(backend, rank=rank, world_size=world)
    return Group(rank, local_rank, world, dist, device)


def proofs_for_rank(total, rank, world):
    """Shard `total` independent proofs over ranks: contiguous blocks, sizes differ by at most one."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def timed_steps(group, step, steps, warmup, device_sync=lambda: None):
    """W untimed steps, then exactly K steps bracketed by (device sync + barrier) on both sides;
    returns the MAX over ranks of the elapsed seconds."""
    for _ in range(warmup):
        step()
    device_sync()
    group.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    device_sync()
    group.barrier()
    return group.max_over_ranks(time.perf_counter() - t0)


def shared_circuit(group, build, path, seed):
    """Several ranks on one node, one circuit: rank 0 calls build() (the expensive host-side circuit construction) and writes the
    result ONCE as a circuit hand-off file (include/glp.h glp_circuit_file_*) at `path`; the other ranks map that file instead of
    repeating the build.  Every rank still proves a witness of its own: NoopGate (padding) rows constrain nothing, so ranks > 0 put
    rank-specific values (SplitMix64 of seed + 1000 rank) into the advice wires of those rows -- a valid witness of the same circuit,
    a different proof.  Returns (desc, circuit_file_or_None, seconds rank 0 spent in build()); the caller closes the file after use.
    The file's name is removed as soon as every rank has mapped it."""
    import numpy as np
    from . import binding as glp
    cf, t_build = None, 0.0
    if group.rank == 0:
        import atexit
        t0 = time.perf_counter()
        desc = build()
        t_build = time.perf_counter() - t0
        atexit.register(lambda: os.path.exists(path) and os.unlink(path))          # also on an early exit
        glp.write_circuit_file(path, desc, with_witness=True)
    group.barrier()                                                             # the file is complete
    if group.rank != 0:
        cf = glp.CircuitFile(path, verify_checksum=False)      # mapped; the canonical-form scan of every section still runs
        desc = cf.desc
        w = np.array(desc.wires)
        noops = [i for i, g in enumerate(desc.gates) if g["type"] == 0]
        nadv = desc.num_wires - desc.num_routed_wires
        if noops and nadv:
            pad = np.nonzero(desc.constants[desc.gates[noops[0]]["selector_index"]] == np.uint64(noops[0]))[0]
            if len(pad):
                w[desc.num_routed_wires:, pad] = glp.splitmix_field(seed + 1000 * group.rank, nadv * len(pad)).reshape(nadv, len(pad))
        desc.wires = w
    group.barrier()                                                             # every rank has mapped it: the name can go
    if group.rank == 0 and os.path.exists(path):
        os.unlink(path)
    return desc, cf, t_build
