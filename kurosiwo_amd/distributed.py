"""Process-level data parallelism of the trainers: one process per GPU, launched by
`python -m torch.distributed.run --nproc-per-node N main.py ...` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment);
backend "nccl" (= RCCL over xGMI on ROCm) on GPUs, "gloo" for the CPU tests.

The reference is single-process (training/change_detection_trainer.py, utilities/utilities.py:96-103: one shuffled DataLoader with
drop_last).  Here every rank builds the SAME loader (same seed => same shuffle order), takes the contiguous shard
[r*B/W, (r+1)*B/W) of each global batch, trains on it with the fused step (gradient buckets all-reduced during backward,
kurosiwo_amd/dp.py; 1/W folded into the optimiser kernel), and evaluates its shard of every validation batch; the 4x4 confusion
matrices and loss sums are all-reduced, so every rank returns the same metrics; only rank 0 prints and writes checkpoints.
BatchNorm statistics stay per rank (the reference has no SyncBN): results equal a single-GPU run at batch B/W per BN call.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1"))


def set_hw_queues():
    """Data-parallel runs add two streams to the three of the train step (the bucket issue stream and ProcessGroupNCCL's own).  The HIP
    runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with five streams on four queues two streams of the step
    share a queue and lose their concurrency -- measured on MI355X, SNUNet bs 32, ONE-rank RCCL (no collective in flight, so a guess
    until two ranks have run): +7 % step time at 4 or 5 queues, +2 % at 6 or 7, +14 % at 8 (profiles/r05_dp_hw_queues.txt).
    Called by the entry points (main.py through init_distributed, bench.py), not at package import: it only defaults the variable
    (an explicit setting wins), and says so when the HIP runtime is already up and the setting can no longer take effect."""
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        import warnings
        warnings.warn("kurosiwo_amd: the HIP runtime was initialised before init_distributed(); GPU_MAX_HW_QUEUES=7 (five streams under "
                      "data parallelism) cannot be applied any more -- export it in the launcher's environment")
        return
    os.environ["GPU_MAX_HW_QUEUES"] = "7"


def init_distributed(configs=None):
    """Join the process group described by the torchrun environment (no-op for a single process).  Returns (rank, local_rank, world)
    and, when `configs` is given, records them and points configs['device'] at this rank's GPU."""
    world = env_world()
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    # KSMI_DP_FORCE=1: join a ONE-rank group as well, so that the collective path (RCCL communicator, device_id binding, bucket hooks,
    # stream joins) runs on a single GPU -- tests/test_gpu_dp.py::test_main_entry_on_the_rccl_backend_one_rank
    if (world > 1 or os.environ.get("KSMI_DP_FORCE")) and not dist.is_initialized():
        set_hw_queues()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # KSMI_DIST_BACKEND=gloo: several ranks on ONE GPU (the single-GPU test box; RCCL refuses duplicate devices)
        backend = os.environ.get("KSMI_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local = local % torch.cuda.device_count() if backend == "gloo" else local
            torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if configs is not None:
        configs["rank"], configs["local_rank"], configs["world_size"] = rank, local, world
        if world > 1 and torch.cuda.is_available():
            configs["gpu"] = local
            configs["device"] = f"cuda:{local}"
    return rank, local, world


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_initialized() else 0


def is_main():
    return get_rank() == 0


def shard_batch(batch, rank=None, world=None, even=True):
    """Contiguous rank shard of a collated batch tuple (tensors sliced on dim 0; the lists of per-channel scale tensors of
    dataset/Dataset.py:824-860 element-wise).  even=True (training: loaders use drop_last): the global batch must divide by the
    world size; even=False (evaluation: the last batch is ragged): rank r takes [n*r//W, n*(r+1)//W), possibly empty."""
    rank = get_rank() if rank is None else rank
    world = world_size() if world is None else world
    if world == 1 or type(batch).__name__ == "ShardedBatch":        # (dataset.TileBatchLoader reads only this rank's slice)
        return batch

    nb = next((t.shape[0] for t in batch if torch.is_tensor(t) and t.dim() > 0), None)     # samples in the global batch

    def cut(t):
        if torch.is_tensor(t):
            if t.dim() == 0:
                return t
            n = t.shape[0]
            if even and n % world:
                raise ValueError(f"global batch {n} is not divisible by world size {world}")
            return t[n * rank // world:n * (rank + 1) // world]
        if isinstance(t, (list, tuple)):
            if nb is not None and len(t) == nb and not any(torch.is_tensor(x) or isinstance(x, (list, tuple)) for x in t):
                return type(t)(t[nb * rank // world:nb * (rank + 1) // world])      # per-sample python values (names, ids)
            return type(t)(cut(x) for x in t)
        return t
    return tuple(cut(t) for t in batch)


class RankShardBatchSampler(torch.utils.data.Sampler):
    """batch_sampler of the DataLoader route under data parallelism: every rank draws the SAME global order (one permutation per
    epoch seeded with seed + epoch when shuffling; the trainers call set_epoch(epoch) -- see set_loader_epoch -- so a resumed run
    continues with the permutation of its epoch; a world-1 run keeps the reference's DataLoader(shuffle=True) on torch's global
    generator, whose order is a different one) and loads ONLY its contiguous slice
    [r*B/W, (r+1)*B/W) of every global batch, instead of every rank decoding all B samples and distributed.shard_batch throwing
    (W-1)/W of them away.  drop_last as utilities/utilities.py:96-103 (train); a ragged last batch (evaluation) is cut the way
    shard_batch(even=False) cuts it: rank r takes [n*r//W, n*(r+1)//W), possibly empty (the loader then skips nothing: an empty
    slice still yields an empty list so that every rank runs the same number of iterations)."""

    def __init__(self, n, batch_size, shuffle, drop_last, rank, world, seed=999):
        self.n, self.bs, self.shuffle, self.drop_last = int(n), int(batch_size), bool(shuffle), bool(drop_last)
        self.rank, self.world, self.seed, self.epoch = int(rank), int(world), int(seed), 0
        if self.drop_last and self.bs % self.world:
            raise ValueError(f"global batch {self.bs} is not divisible by world size {self.world}")

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return self.n // self.bs if self.drop_last else -(-self.n // self.bs)

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
            self.epoch += 1
        else:
            order = list(range(self.n))
        for b in range(len(self)):
            idx = order[b * self.bs:(b + 1) * self.bs]
            m = len(idx)
            yield idx[m * self.rank // self.world:m * (self.rank + 1) // self.world]


def set_loader_epoch(loader, epoch):
    """tell a rank-sharded loader which epoch's permutation to draw (no-op for loaders without a RankShardBatchSampler)"""
    bs = getattr(loader, "batch_sampler", None)
    if isinstance(bs, RankShardBatchSampler):
        bs.set_epoch(epoch)


def sharded_collate(samples):
    """default_collate of this rank's slice, tagged so that shard_batch passes it through"""
    from .dataset import ShardedBatch
    if not samples:
        return ShardedBatch(())
    return ShardedBatch(torch.utils.data.default_collate(samples))


def make_loader(dataset, batch_size, shuffle, drop_last, num_workers=0, seed=999, pin_memory=True):
    """the reference's DataLoader (utilities/utilities.py:96-121) for one process; under data parallelism the rank-strided form"""
    world = world_size()
    if world == 1:
        return torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers,
                                           pin_memory=pin_memory, drop_last=drop_last)
    sampler = RankShardBatchSampler(len(dataset), batch_size, shuffle, drop_last, get_rank(), world, seed)
    return torch.utils.data.DataLoader(dataset, batch_sampler=sampler, num_workers=num_workers, pin_memory=pin_memory,
                                       collate_fn=sharded_collate)


def all_reduce_sum_(*tensors):
    """In-place SUM over the ranks (confusion matrices, loss / sample counters)."""
    if world_size() > 1:
        for t in tensors:
            if dist.get_backend() == "gloo" and t.is_cuda:          # (CPU-backend tests with device tensors)
                c = t.cpu()
                dist.all_reduce(c, op=dist.ReduceOp.SUM)
                t.copy_(c)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tensors


def broadcast_model_(model, src=0):
    """Rank `src`'s parameters, BatchNorm statistics and counters on every rank (the arenas are flat: three broadcasts)."""
    if world_size() > 1:
        for name in ("flat_params", "flat_buffers", "flat_counters"):
            t = getattr(model, name, None)
            if t is not None and t.numel():
                if dist.get_backend() == "gloo" and t.is_cuda:
                    c = t.cpu()
                    dist.broadcast(c, src)
                    t.copy_(c)
                else:
                    dist.broadcast(t, src)
    return model


def barrier():
    if world_size() > 1:
        dist.barrier()


def broadcast_object(obj, src=0):
    """rank `src`'s picklable object on every rank (e.g. the time-stamped checkpoint directory name)."""
    if world_size() > 1:
        box = [obj]
        dist.broadcast_object_list(box, src=src)
        return box[0]
    return obj
