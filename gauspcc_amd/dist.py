"""Multi-GPU harness for the path: independent scenes shard across ranks (SURVEY.md
section 8e), nothing is exchanged on the data path, and one all_gather collates the
per-scene bitstream statistics.  Works over any torch.distributed backend: RCCL
("nccl") on the MI355X node, gloo in the CPU tests.
"""
import os
import socket
import subprocess
import sys
from dataclasses import dataclass

import torch

STAT_FIELDS = ("num_points", "num_bytes", "enc_s", "dec_s", "coded_nodes", "conv_pairs", "levels", "status")


@dataclass
class SceneStats:
    num_points: int = 0
    num_bytes: int = 0
    enc_s: float = 0.0
    dec_s: float = 0.0
    coded_nodes: int = 0
    conv_pairs: int = 0
    levels: int = 0
    status: int = 0

    def to_tensor(self, device) -> torch.Tensor:
        return torch.tensor([float(getattr(self, f)) for f in STAT_FIELDS], dtype=torch.float64, device=device)

    @classmethod
    def from_row(cls, row) -> "SceneStats":
        kw = {}
        for f, v in zip(STAT_FIELDS, row):
            kw[f] = float(v) if f in ("enc_s", "dec_s") else int(round(float(v)))
        return cls(**kw)


def scenes_for_rank(n_scenes: int, rank: int, world: int):
    """Scene i -> rank i mod world (each rank owns its model replica and its output files)."""
    return list(range(rank, n_scenes, world))


def scene_seed(base_seed: int, scene: int) -> int:
    return base_seed + scene


def collate_stats(local, device, group=None):
    """all_gather the per-scene records of every rank; returns the flat list ordered by
    (rank, local index).  `local` is a list of SceneStats (may be empty on some ranks)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return list(local)
    world = dist.get_world_size(group)
    count = torch.tensor([len(local)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    cmax = max(int(c.item()) for c in counts)
    buf = torch.zeros((max(cmax, 1), len(STAT_FIELDS)), dtype=torch.float64, device=device)
    for i, s in enumerate(local):
        buf[i] = s.to_tensor(device)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    out = []
    for r in range(world):
        rows = bufs[r].cpu().numpy()
        for i in range(int(counts[r].item())):
            out.append(SceneStats.from_row(rows[i]))
    return out


def max_over_ranks(seconds: float, device, group=None) -> float:
    import torch.distributed as dist

    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def rank_cpu_set(local_rank: int, local_world: int, allowed=None):
    """The host threads of rank `local_rank` of `local_world` ranks on this node: an even, contiguous share of the CPUs this process may
    use (sorted ids, so that on the usual numbering a share stays inside one socket / NUMA node).  Eight ranks each start parser, writer
    and OpenMP threads (gpcc_write_files / gpcc_read_files run 8 native threads per call); unpinned they pile onto the cores of one
    NUMA node while the others idle.  Never returns an empty set: with fewer CPUs than ranks the shares wrap around."""
    cpus = sorted(os.sched_getaffinity(0) if allowed is None else allowed)
    local_world = max(1, int(local_world))
    local_rank = int(local_rank) % local_world
    if len(cpus) < local_world:
        return {cpus[local_rank % len(cpus)]}
    per = len(cpus) // local_world
    return set(cpus[local_rank * per:(local_rank + 1) * per])


def pin_rank_threads(local_rank=None, local_world=None) -> int:
    """Give this rank its share of the node's host threads (rank_cpu_set) -- call it BEFORE anything touches the GPU or starts a thread
    pool, so that every thread the process creates later inherits the mask.  Reads LOCAL_RANK / LOCAL_WORLD_SIZE (torch.distributed.run)
    when not given; a single rank is left alone.  Returns the number of CPUs in the mask now in force.  GAUSPCC_NO_PIN=1 disables it."""
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if local_world <= 1 or os.environ.get("GAUSPCC_NO_PIN") == "1" or not hasattr(os, "sched_setaffinity"):
        return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    mine = rank_cpu_set(local_rank, local_world)
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        pass
    n = len(os.sched_getaffinity(0))
    os.environ["OMP_NUM_THREADS"] = str(max(1, min(int(os.environ.get("OMP_NUM_THREADS", n)), n)))
    return n


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(script: str, argv, nproc: int, port: int = 0, module: bool = False) -> int:
    """Start `nproc` ranks of `script` on this node, one per GPU, the way the driver does:
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P script argv`
    (module=True: `script` is a module name, started as `... --module script argv`).
    Called by a parent that has NOT touched the GPU (an exec / fork from a process with an initialised HIP runtime takes the
    box down); the children are ordinary subprocesses whose stdout / stderr pass straight through (rank 0 prints the JSON
    line), and the launcher's exit code is returned."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port or free_port())] + (["--module"] if module else []) + [script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool: RCCL needs it
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def init_from_env(prefer_gpu: bool = True):
    """Process group of a rank started by torch.distributed.run (RANK / WORLD_SIZE / MASTER_* in the environment): RCCL
    ("nccl") with the rank's own GPU, gloo without one.  Returns (rank, world, device); (0, 1, device) outside a launcher."""
    import torch.distributed as dist

    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    pin_rank_threads(local)          # before the first GPU call: every later thread inherits the rank's share of the host cores
    gpu = prefer_gpu and torch.cuda.is_available()
    device = torch.device("cuda", local) if gpu else torch.device("cpu")
    if gpu:
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        if gpu:
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    return rank, world, device


def scene_order(n_scenes: int, world: int):
    """collate_stats returns records ordered by (rank, local index); scene_order(n, world)[p] is the scene of position p."""
    return [s for r in range(world) for s in scenes_for_rank(n_scenes, r, world)]
