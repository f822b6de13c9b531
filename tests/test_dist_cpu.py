"""world_size-2 gloo test of the N>1 path: scene sharding + stats all_gather + MAX timing
(the same code bench.py runs over RCCL)."""
import json
import os
import socket
import subprocess
import sys

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gauspcc_amd.dist import SceneStats, collate_stats, max_over_ranks, scene_seed, scenes_for_rank

    scenes = scenes_for_rank(5, rank, world)
    local = [SceneStats(num_points=1000 + s, num_bytes=10 * s + rank, enc_s=0.5 + s, dec_s=0.25 * s,
                        coded_nodes=scene_seed(1234, s), conv_pairs=2 ** 40 + s, levels=10 + s, status=0) for s in scenes]
    allst = collate_stats(local, torch.device("cpu"))
    tmax = max_over_ranks(1.0 + rank, torch.device("cpu"))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, scenes, [(s.num_points, s.num_bytes, s.enc_s, s.coded_nodes, s.conv_pairs, s.levels) for s in allst], tmax))


def test_scene_sharding_and_stats_allgather_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    assert res[0][2] == res[1][2]                      # every rank sees the same collated table
    pts = [r[0] for r in res[0][2]]
    assert pts == [1000, 1002, 1004, 1001, 1003]       # ordered by (rank, local index)
    assert [r[4] for r in res[0][2]][0] == 2 ** 40     # int64-sized counters survive the float64 transport
    assert res[0][3] == res[1][3] == 2.0               # MAX over ranks


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two ranks through torch.distributed.run
    (before touching any GPU), relays rank 0's JSON line and the exit code.  Here on gloo, without a GPU."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--points", "1000", "--selftest-launcher"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout     # one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["seeds"] == [1234, 1235] and out["max_elapsed"] == 2.0


def test_bench_relays_a_failing_rank():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    # without --selftest-launcher the ranks need a GPU: on this box they exit non-zero, and so must the parent
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--points", "1000", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode != 0


def test_batch_cli_two_ranks_write_the_csv_of_one_rank(tmp_path):
    """BASELINE configs[3] entry point (reference batch driver: compress_ue_4stage_conv.py:56-62,245-286): the compress CLI
    under two gloo ranks -- file i on rank i mod 2, per-file rows through dist.collate_stats, rank 0 writes the one CSV --
    gives the CSV (input order, avg row) and the .bin files of a single process.  Stub codec: no GPU here."""
    import numpy as np
    import pandas as pd

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.RandomState(5)
    src = tmp_path / "in"
    src.mkdir()
    for i in range(5):
        np.save(src / f"c{i}.npy", rng.randint(0, 512, size=(300 + 37 * i, 3)).astype(np.float64))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    out = {}
    for tag, extra in (("one", []), ("two", ["--gpus", "2"])):
        cmd = [sys.executable, "-m", "gauspcc_amd.cli.compress", "--input_glob", str(src), "--output_folder", str(tmp_path / tag / "bin"),
               "--resultdir", str(tmp_path / tag / "res"), "--ckpt", "synthetic", "--selftest-stub"] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-3000:]
        assert r.stdout.count("Total: 5 |") == 1, r.stdout        # one summary line, from rank 0
        out[tag] = pd.read_csv(tmp_path / tag / "res" / "ue_4stage_conv_data5.csv")
    a, b = out["one"], out["two"]
    assert a["filedir"].tolist() == [f"c{i}.npy" for i in range(5)] + ["avg"] == b["filedir"].tolist()
    for col in ("bpp", "enc_time", "file_size_bits", "num_points"):
        assert np.allclose(a[col].to_numpy(), b[col].to_numpy(), rtol=1e-12), col
    for i in range(5):
        assert (tmp_path / "one" / "bin" / f"c{i}.npy.bin").read_bytes() == (tmp_path / "two" / "bin" / f"c{i}.npy.bin").read_bytes()


def test_batch_cli_jobs_and_decompress_under_two_ranks(tmp_path):
    """`--jobs` is honoured per rank under `--gpus N` (round 3 silently ignored it), and the decompress CLI shards like the
    compressor (reference batch driver: decompress_ue_4stage_conv.py:46-192): two gloo ranks, two files in flight each,
    give the CSV / .bin / .ply files and the one summary line of a single process.  Stub codec: no GPU here."""
    import numpy as np
    import pandas as pd

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.RandomState(11)
    src = tmp_path / "in"
    src.mkdir()
    for i in range(7):
        np.save(src / f"c{i}.npy", rng.randint(0, 512, size=(200 + 41 * i, 3)).astype(np.float64))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    csv, summary = {}, {}
    for tag, extra in (("one", []), ("two", ["--gpus", "2", "--jobs", "2"])):
        cmd = [sys.executable, "-m", "gauspcc_amd.cli.compress", "--input_glob", str(src), "--output_folder", str(tmp_path / tag / "bin"),
               "--resultdir", str(tmp_path / tag / "res"), "--ckpt", "synthetic", "--selftest-stub"] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-3000:]
        assert r.stdout.count("Total: 7 |") == 1, r.stdout
        if extra:
            assert "2 files in flight per rank" in r.stdout, r.stdout
        csv[tag] = pd.read_csv(tmp_path / tag / "res" / "ue_4stage_conv_data7.csv")
        cmd = [sys.executable, "-m", "gauspcc_amd.cli.decompress", "--input_glob", str(tmp_path / tag / "bin" / "*.bin"),
               "--output_folder", str(tmp_path / tag / "ply"), "--ckpt", "synthetic", "--selftest-stub"] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("Total: 7 |")]
        assert len(lines) == 1, r.stdout                          # one summary line, from rank 0, over all seven files
        summary[tag] = lines[0].split("|")[1].strip()             # "Decoding time:0.00Xs": the mean over every rank's files
    assert csv["one"]["filedir"].tolist() == csv["two"]["filedir"].tolist()
    for col in ("bpp", "enc_time", "file_size_bits", "num_points"):
        assert np.allclose(csv["one"][col].to_numpy(), csv["two"][col].to_numpy(), rtol=1e-12), col
    assert summary["one"] == summary["two"]
    for i in range(7):
        for sub, ext in (("bin", ".npy.bin"), ("ply", ".npy.bin.ply")):
            assert (tmp_path / "one" / sub / f"c{i}{ext}").read_bytes() == (tmp_path / "two" / sub / f"c{i}{ext}").read_bytes()


def _bench_selftest(extra, timeout=600):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["OMP_NUM_THREADS"] = "1"
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--points", "1000", "--selftest-launcher"] + extra,
                          capture_output=True, text=True, timeout=timeout, env=env, cwd=root)


def test_eight_ranks_uneven_batch_and_disjoint_host_threads():
    """The shape of the first 8-GPU box, pinned on gloo (scene i -> rank i mod 8, one all_gather of the stats, MAX of the
    times): `bench.py --gpus 8 --scenes-per-gpu 3` starts its own eight ranks; a batch of 21 scenes (not a multiple of 8) leaves ranks
    5-7 one scene short; every scene is collated exactly once, in (rank, local index) order; and every rank has pinned itself to its own
    share of the host cores before anything else (dist.pin_rank_threads) -- disjoint where the box has at least eight."""
    r = _bench_selftest(["--gpus", "8", "--scenes-per-gpu", "3"])
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8 and out["scenes_seen"] == 24
    assert out["seeds"] == [1234 + s for s in out["scene_order"]] and sorted(out["scene_order"]) == list(range(24))
    assert out["max_elapsed"] == 8.0
    r = _bench_selftest(["--gpus", "8", "--selftest-scenes", "21"])
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["scenes_seen"] == 21 and sorted(out["scene_order"]) == list(range(21))
    assert out["scene_order"][:3] == [0, 8, 16] and out["scene_order"][-2:] == [7, 15]           # rank 0 has three scenes, rank 7 two
    assert out["seeds"] == [1234 + s for s in out["scene_order"]]
    cpus = out["rank_cpus"]                                                                      # [count, first id, last id] per rank
    assert len(cpus) == 8 and all(c[0] >= 1 for c in cpus)
    have = len(os.sched_getaffinity(0))
    if have >= 8:
        assert all(c[0] == have // 8 for c in cpus)
        spans = sorted((c[1], c[2]) for c in cpus)
        assert all(a[1] < b[0] for a, b in zip(spans, spans[1:])), spans                          # disjoint, contiguous shares


def test_eight_ranks_one_failing_rank_ends_the_job():
    """A rank that dies before the rendezvous (a GPU that does not come up): the launcher ends the other seven and `bench.py --gpus 8` relays a
    non-zero exit code instead of hanging in the rendezvous or printing a line for a partial job."""
    r = _bench_selftest(["--gpus", "8", "--selftest-fail-rank", "5"], timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout


def test_rank_cpu_sets_partition_the_allowed_cpus():
    from gauspcc_amd.dist import rank_cpu_set

    allowed = set(range(4, 68))                                         # e.g. a cgroup's 64 CPUs
    shares = [rank_cpu_set(r, 8, allowed) for r in range(8)]
    assert all(len(s) == 8 for s in shares) and set().union(*shares) == allowed
    assert all(a.isdisjoint(b) for i, a in enumerate(shares) for b in shares[i + 1:])
    assert shares[0] == set(range(4, 12)) and shares[7] == set(range(60, 68))
    few = [rank_cpu_set(r, 8, {0, 1, 2}) for r in range(8)]             # fewer CPUs than ranks: one each, wrapping
    assert all(len(s) == 1 for s in few) and set().union(*few) == {0, 1, 2}
    assert rank_cpu_set(0, 1, allowed) == allowed
