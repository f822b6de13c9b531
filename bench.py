#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its configs[1]: Mpoints/s for GausPcgc
encode + decode of a 1M-anchor synthetic cloud on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one scene: gpcc_encode (points resident in
HBM -> container bytes on the host) followed by gpcc_decode (container bytes -> points
in HBM).  Scenes are independent, so with N GPUs every rank codes its own scene (weak
scaling, no data-path collective); one RCCL all_gather collates the bitstream stats.
Weights are seeded synthetic (the reference ships no checkpoint) and the cloud comes
from the in-repo counter-based generator -- "data": "synthetic".

One JSON line on stdout (rank 0).  Besides the contract keys it carries
  roofline      the dominant kernel (k_sparse_conv, fp32 MFMA) timed live with HIP
                events on its own stream inside libgauspcc (gpcc_profile_*)
  cpu_baseline  the oracle (CPU restatement of the reference algorithm) on a bounded sample
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The HIP runtime multiplexes a process's streams onto 4 hardware queues by default; a context uses three streams, so two
# scenes in flight (the `scenes_in_flight` pass) would share queues and serialise.  No effect on the timed one-scene region.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

MFMA_F32_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 matrix peak
HBM_PEAK_GBPS = 8000.0        # same guide: HBM3E 8.0 TB/s spec (6.3 TB/s measured with a float4 copy)


def launcher_selftest(args, rank, world):
    """The rank-side skeleton of a multi-GPU run without a GPU (gloo): process group from the launcher's environment, the
    stats all_gather and the MAX of the elapsed time, one JSON line from rank 0.  Covers `bench.py --gpus N` starting its
    own ranks (tests/test_dist_cpu.py)."""
    import torch
    import torch.distributed as dist

    from gauspcc_amd.dist import SceneStats, collate_stats, max_over_ranks, scene_seed

    from gauspcc_amd.dist import scene_order, scenes_for_rank

    if args.selftest_fail_rank == rank:
        raise SystemExit(3)          # a rank that dies before the rendezvous: the launcher must end the others and relay a non-zero code
    dist.init_process_group("gloo")
    cpu = torch.device("cpu")
    # the batch as main() shards it: --selftest-scenes S scenes in all (default: scenes_per_gpu x world), scene i on rank i mod world -- an S that
    # is not a multiple of the world size leaves the last ranks one scene short (or with none)
    total = args.selftest_scenes if args.selftest_scenes > 0 else max(1, args.scenes_per_gpu) * world
    mine = [SceneStats(num_points=args.points, num_bytes=100 + s, enc_s=0.01 * (s + 1), dec_s=0.02, coded_nodes=scene_seed(1234, s),
                       conv_pairs=1, levels=3, status=0) for s in scenes_for_rank(total, rank, world)]
    scenes = collate_stats(mine, cpu)
    elapsed = max_over_ranks(1.0 + rank, cpu)
    cpus = sorted(os.sched_getaffinity(0))
    cpu_rows = collate_stats([SceneStats(num_points=len(cpus), num_bytes=cpus[0], coded_nodes=cpus[-1], status=rank)], cpu)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"selftest": "launcher", "n_gpus": world, "ranks_seen": len({s.status for s in cpu_rows}), "scenes_seen": len(scenes),
                          "seeds": [s.coded_nodes for s in scenes], "scene_order": scene_order(total, world),
                          "max_elapsed": elapsed, "value": args.points * len(scenes) / elapsed / 1e6,
                          "rank_cpus": [[s.num_points, s.num_bytes, s.coded_nodes] for s in cpu_rows]}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)    # a step is ~48 ms: 20 timed + 5 warm-up steps are 1.2 s, and the figure no longer rides on the clock ramp of the first steps
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--kernel-size", type=int, default=5)
    ap.add_argument("--chunk-log2", type=int, default=11)
    ap.add_argument("--scenes-per-gpu", type=int, default=1, help="independent scenes each rank codes per step, one after the other (weak scaling: scene i of the "
                    "batch -> rank i mod N, gauspcc_amd.dist.scenes_for_rank; BASELINE configs[3] is --gpus 8 with one or more scenes per GPU)")
    ap.add_argument("--solo-scenes", action="store_true", help="with --scenes-per-gpu K > 1: code a rank's K scenes one after the other (gpcc_encode / gpcc_decode) instead of "
                    "through one chain of launches (gpcc_encode_batch / gpcc_decode_batch, the default: BASELINE configs[3] is a BATCHED encode)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="OpenMP threads of the CPU baseline (0 = the fastest of 8 / 16 / 32 / 64 / 128 / all cores on a short calibration run; a positive value pins the count, capped by the cores this process may use)")
    ap.add_argument("--measure-traffic", action="store_true", help="(informative) leave roofline.traffic null instead of quoting profiles/: run tools/pmc_traffic.sh for a fresh figure")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="points of the CPU-baseline cloud (0 = skip; default: the metric's own 1 M config)")
    ap.add_argument("--skip-v0", action="store_true", help="do not encode the reference-layout container for bytes_v0")
    ap.add_argument("--event-steps", type=int, default=1, help="timed steps that carry the HIP-event brackets around the conv launches (-1 = all; each bracket costs its stream ~5 us, ~0.65 ms per step)")
    ap.add_argument("--scenes-in-flight", type=int, default=2, help="also report the throughput with this many independent scenes in flight on the GPU "
                    "(own context, stream and host thread each; untimed extra pass on rank 0 at N = 1; 0 = skip)")
    ap.add_argument("--side-anchors", type=int, default=200_000, help="anchors of the synthetic HAC-style scene behind the `side_paths` object (SURVEY 8f: attribute "
                    "loop, Gaussian coder, mlp_grid, generate_neural_gaussians + rasteriser, torchac shim; untimed pass on rank 0 at N = 1; 0 = skip)")
    ap.add_argument("--skip-stages", action="store_true", help="do not run the extra pass that times the HBM-bound stages")
    ap.add_argument("--skip-sizes", action="store_true", help="do not run the untimed passes behind `sizes` (one scene of 10 k / 100 k / 1 M points) and `batched` "
                    "(K scenes through one chain of launches: gpcc_encode_batch / gpcc_decode_batch)")
    ap.add_argument("--selftest-launcher", action="store_true", help=argparse.SUPPRESS)   # tests/test_dist_cpu.py: the N > 1 launch path on gloo, no GPU
    ap.add_argument("--selftest-scenes", type=int, default=0, help=argparse.SUPPRESS)     # (selftest) scenes of the batch in all; 0 = scenes_per_gpu x ranks
    ap.add_argument("--selftest-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)  # (selftest) this rank exits before the rendezvous
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU), BEFORE anything
        # here touches the GPU; this parent only relays the ranks' output (rank 0's JSON line) and their exit code
        from gauspcc_amd.dist import launch_ranks

        sys.exit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    # each rank keeps to its own share of the host cores (parser / writer / OpenMP threads): before anything touches the GPU
    from gauspcc_amd.dist import pin_rank_threads

    pin_rank_threads(local_rank)
    if args.selftest_launcher:
        return launcher_selftest(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:  # under torch.distributed.run, also for a single rank
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=device)  # RCCL over xGMI

    from gauspcc_amd import _lib, runtime
    from gauspcc_amd.dist import SceneStats, collate_stats, max_over_ranks, scene_seed, scenes_for_rank
    from gauspcc_amd.pcc_utils import _decode_batch, _decode_bytes, _encode_batch, _encode_view
    from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict

    k = args.kernel_size
    sd = synthetic_state_dict(32, k)
    model = runtime.Model(sd, 32, k, local_rank)
    # the batch: scenes_per_gpu x world independent scenes, scene i on rank i mod world (scene i of rank r at K = 1 is scene r)
    K = max(1, args.scenes_per_gpu)
    my_scenes = scenes_for_rank(K * world, rank, world)
    seed = scene_seed(1234, my_scenes[0])
    clouds = [synthetic_cloud(args.points, seed=scene_seed(1234, s)) for s in my_scenes]
    xs_batch = [torch.tensor(c, device=device) for c in clouds]  # inputs resident in HBM before the timed region
    pts, x = clouds[0], xs_batch[0]
    ctx = runtime.context(device)
    L = _lib.lib()

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(device)

    batch_mode = K > 1 and not args.solo_scenes and args.chunk_log2 != 0

    def step():
        te = td = 0.0
        if batch_mode:
            # the rank's K scenes through ONE chain of launches; `data` / `dec` of the step are scene 0's (checked after the timed region)
            t0 = time.perf_counter()
            views, sts, _ = _encode_batch(xs_batch, model, args.chunk_log2, [1] * K, view=True)
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            decs, _, _, _ = _decode_batch(views, model, device)
            torch.cuda.synchronize(device)
            t2 = time.perf_counter()
            st0 = sts[0]
            st0.coded_nodes = sum(s_.coded_nodes for s_ in sts)
            return views[0], st0, decs[0], t1 - t0, t2 - t1
        for xi in reversed(xs_batch):   # scene 0 last: `data` / `dec` of the step are its (checked after the timed region)
            t0 = time.perf_counter()
            data, st = _encode_view(xi, model, args.chunk_log2, 1)   # the bitstream stays in the library's pinned host buffer
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            dec, _, _ = _decode_bytes(data, model, device)
            torch.cuda.synchronize(device)
            t2 = time.perf_counter()
            te += t1 - t0
            td += t2 - t1
        return data, st, dec, te, td

    for _ in range(args.warmup):
        step()
    ev_steps = args.steps if args.event_steps < 0 else min(args.event_steps, args.steps)
    _lib.check(L.gpcc_profile_enable(ctx, 1 if ev_steps > 0 else 0))
    barrier()
    t_start = time.perf_counter()
    enc_s = dec_s = 0.0
    for i in range(args.steps):
        if i == ev_steps and ev_steps > 0:
            _lib.check(L.gpcc_profile_enable(ctx, 3))   # keep what was collected, stop recording
        data, st, dec, te, td = step()
        enc_s += te
        dec_s += td
    barrier()
    elapsed = time.perf_counter() - t_start
    prof = _lib.Profile()
    _lib.check(L.gpcc_profile_get(ctx, C.byref(prof)))
    _lib.check(L.gpcc_profile_enable(ctx, 0))

    # size of the same cloud in the reference's container layout (one coder stream per level and stage, chunk_log2 = 0):
    # what the chunked default costs in bytes, outside the timed region
    bytes_v0 = None
    if args.chunk_log2 and rank == 0 and not args.skip_v0:
        v0, _ = _encode_view(x, model, 0, 1)
        bytes_v0 = len(v0)
        data, st = _encode_view(x, model, args.chunk_log2, 1)   # the view above shares the context's buffer: restore `data`

    # HBM-bound stages (SURVEY 8d "which roofline"): one more encode + decode with every stage bracketed by HIP events on the
    # stream it runs on (gpcc_profile_enable(ctx, 2)) -- behind the timed region, because the brackets cost stream time
    stages = []
    if rank == 0 and not args.skip_stages:
        _lib.check(L.gpcc_profile_enable(ctx, 2))
        d2, _ = _encode_view(x, model, args.chunk_log2, 1)
        _decode_bytes(d2, model, device)
        torch.cuda.synchronize(device)
        arr = (_lib.Stage * 8)()
        ns = C.c_int()
        _lib.check(L.gpcc_profile_stages(ctx, arr, 8, C.byref(ns)))
        _lib.check(L.gpcc_profile_enable(ctx, 0))
        for i in range(ns.value):
            ms, by = arr[i].ms, arr[i].bytes
            stages.append({"stage": arr[i].name.decode(), "bound": "hbm", "ms_per_step": round(ms, 3), "algorithmic_bytes": int(by),
                           "achieved": round(by / (ms * 1e-3) / 1e9, 1) if ms > 0 else 0.0, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if ms > 0 else 0.0, "event_brackets": int(arr[i].brackets),
                           # ms_per_step is the bracketed stream time (the octree / tile-list brackets of a decode run BESIDE the parent trunk's
                           # convolutions); critical_ms is the part of it during which no convolution bracket was open: what the step waits for
                           "critical_ms": round(arr[i].critical_ms, 3)})
        data, st = _encode_view(x, model, args.chunk_log2, 1)   # `data` is a view of the context's buffer: restore it

    # Throughput mode (informative, not `value`): S independent scenes in flight on this GPU, each with its own context,
    # stream and host thread -- what one scene leaves idle (launch tails, the range decoder's serial stretches, the small
    # levels) another fills.  Behind the timed region; per-scene latency grows.
    inflight = None
    if rank == 0 and world == 1 and args.scenes_in_flight > 1:
        import threading

        S = args.scenes_in_flight
        xs = [x] + [torch.tensor(synthetic_cloud(args.points, seed=scene_seed(1234, 100 + i)), device=device) for i in range(1, S)]
        ctxs = [ctx]   # scene 0 keeps the context of the timed region
        for _ in range(1, S):
            h = C.c_void_p()
            _lib.check(L.gpcc_ctx_create(local_rank, C.byref(h)))
            ctxs.append(h)
        streams = [torch.cuda.Stream(device=device) for _ in range(S)]
        gate = threading.Barrier(S)
        took = [0.0] * S
        fsteps = max(4, min(2 * args.steps, 10))
        stagger = (enc_s + dec_s) / args.steps / S   # scenes that start in lockstep stay in lockstep (both in their convolutions
        # at once: no gain at all); requests of a real server arrive out of phase, so scene i starts i / S of a step late

        failed = []

        def scene(i):
            try:
                scene_body(i)
            except BaseException as e:   # a thread's exception would otherwise vanish and leave its `took` at 0: an inflated figure
                failed.append((i, repr(e)))
                try:
                    gate.abort()
                except Exception:
                    pass

        def scene_body(i):
            def one():
                pb, nb, s1 = C.c_void_p(), C.c_int64(), _lib.Stats()
                sp = C.c_void_p(streams[i].cuda_stream)
                _lib.check(L.gpcc_encode(ctxs[i], model.handle, xs[i].data_ptr(), xs[i].shape[0], args.chunk_log2, runtime.f16_bits(1), C.byref(pb), C.byref(nb),
                                         C.byref(s1), sp))
                px, nn, pq, s2 = C.c_void_p(), C.c_int64(), C.c_uint16(), _lib.Stats()
                _lib.check(L.gpcc_decode(ctxs[i], model.handle, pb, nb.value, C.byref(px), C.byref(nn), C.byref(pq), C.byref(s2), sp))
                assert nn.value == xs[i].shape[0]
            one()
            gate.wait()
            t0 = time.perf_counter()
            time.sleep(i * stagger)
            for _ in range(fsteps):
                one()
            streams[i].synchronize()
            took[i] = time.perf_counter() - t0   # includes the stagger

        torch.cuda.synchronize(device)
        th = [threading.Thread(target=scene, args=(i,)) for i in range(S)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for h in ctxs[1:]:
            L.gpcc_ctx_destroy(h)
        if failed:
            raise SystemExit(f"scenes in flight: {failed}")
        el = max(took)
        inflight = {"scenes": S, "value": round(S * fsteps * args.points / el / 1e6, 4), "unit": "Mpoints/s", "steps": fsteps,
                    "ms_per_scene_step": round(1e3 * el / fsteps, 3)}

    # Scenes of realistic size, and batches of them (untimed passes on rank 0 at N = 1; `value` stays the one-scene 1 M figure).
    # sizes:   one scene at a time (gpcc_encode / gpcc_decode), per size enc / dec ms, Mpoints/s and kernel launches per call.
    # batched: K scenes through ONE chain of launches (gpcc_encode_batch / gpcc_decode_batch, csrc/forest.hpp; BASELINE configs[3] is a
    #          batched encode): Mpoints/s over all K scenes, launches per call, every scene's bytes compared with its solo encode.
    sizes, batched = None, None
    if rank == 0 and world == 1 and not args.skip_sizes:
        from gauspcc_amd.pcc_utils import _decode_batch, _encode_batch

        def timed(fn, reps):
            fn()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize(device)
            return (time.perf_counter() - t0) / reps

        def launches(fn):
            L.gpcc_debug_launches(1)
            fn()
            return int(L.gpcc_debug_launches(1))

        sizes = []
        solo_bytes = {}
        for n_s in (10_000, 100_000, 1_000_000):
            xs_s = x if n_s == args.points else torch.tensor(synthetic_cloud(n_s, seed=scene_seed(1234, 200)), device=device)
            reps = 20 if n_s < 1_000_000 else 5
            blob = bytes(_encode_view(xs_s, model, args.chunk_log2, 1)[0])
            solo_bytes[n_s] = len(blob)
            te_s = timed(lambda: (_encode_view(xs_s, model, args.chunk_log2, 1), torch.cuda.synchronize(device)), reps)
            td_s = timed(lambda: (_decode_bytes(blob, model, device), torch.cuda.synchronize(device)), reps)
            sizes.append({"points": n_s, "enc_ms": round(te_s * 1e3, 3), "dec_ms": round(td_s * 1e3, 3), "value": round(n_s / (te_s + td_s) / 1e6, 4), "unit": "Mpoints/s",
                          "kernels_per_encode": launches(lambda: _encode_view(xs_s, model, args.chunk_log2, 1)),
                          "kernels_per_decode": launches(lambda: _decode_bytes(blob, model, device))})
        batched = []
        for K_b, n_b in ((8, 100_000), (32, 10_000), (2, 1_000_000)):
            xs_b = [torch.tensor(synthetic_cloud(n_b, seed=scene_seed(1234, 300 + i)), device=device) for i in range(K_b)]
            reps = 10 if K_b * n_b < 2_000_000 else 4
            views, _, was_b = _encode_batch(xs_b, model, args.chunk_log2, [1] * K_b, view=True)
            blobs_b = [bytes(v) for v in views]
            same = all(bytes(_encode_view(xi, model, args.chunk_log2, 1)[0]) == bi for xi, bi in zip(xs_b, blobs_b))
            te_b = timed(lambda: (_encode_batch(xs_b, model, args.chunk_log2, [1] * K_b, view=True), torch.cuda.synchronize(device)), reps)
            td_b = timed(lambda: (_decode_batch(blobs_b, model, device), torch.cuda.synchronize(device)), reps)
            outs_b, _, _, was_db = _decode_batch(blobs_b, model, device)
            rt = all(torch.equal(o, _decode_bytes(bi, model, device)[0]) for o, bi in zip(outs_b, blobs_b))
            batched.append({"scenes": K_b, "points_per_scene": n_b, "enc_ms": round(te_b * 1e3, 3), "dec_ms": round(td_b * 1e3, 3),
                            "value": round(K_b * n_b / (te_b + td_b) / 1e6, 4), "unit": "Mpoints/s",
                            "kernels_per_encode": launches(lambda: _encode_batch(xs_b, model, args.chunk_log2, [1] * K_b, view=True)),
                            "kernels_per_decode": launches(lambda: _decode_batch(blobs_b, model, device)),
                            "one_tree": bool(was_b and was_db), "bytes_identical_to_solo": bool(same), "decode_identical_to_solo": bool(rt)})
            del xs_b
        data, st = _encode_view(x, model, args.chunk_log2, 1)   # `data` is a view of the context's buffer: restore it

    # The reference's own container layout, timed: chunk_log2 = 0 writes one torchac-compatible stream per level
    # and stage (pcc_utils.py:174-177, layout :198-203) -- the only layout the reference can read.  Every stream is ONE dependent chain, so
    # the CODER of this layout runs on the host (csrc/hostcoder.hpp: an encode's streams on a pool of native threads, a decode's -- sequentially
    # dependent -- on one), the network on the device; untimed pass behind the timed region, `value` stays the chunked container's figure.
    # low_rate (item 3b): the same codec at a realistic rate -- (i) the bench cloud under the `peaky` model (synth.peaky_state_dict: head biases
    # = log of the stage symbols' empirical frequencies, i.e. the context-free entropy of the occupancy symbols; 3-6 bpp on this cloud needs
    # spatial context, i.e. a trained checkpoint), (ii) a SOLID cloud (synth.solid_cloud) under its own peaky model: 2.5 bits per coded node,
    # the range coder's low-entropy regime.  chunk_overhead_frac is measured (chunked bytes against chunk_log2 = 0 bytes), not extrapolated.
    reference_layout, low_rate, chunk_sweep = None, None, None
    if rank == 0 and world == 1 and not args.skip_sizes and args.chunk_log2:
        from gauspcc_amd.synth import peaky_state_dict, solid_cloud, stage_symbol_frequencies

        def timed2(fn, reps):
            fn()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize(device)
            return (time.perf_counter() - t0) / reps

        def sorted_rows(a):
            return a[np.lexsort((a[:, 0], a[:, 1], a[:, 2]))]

        def layout_point(xs_l, pts_l, model_l, clog, reps):
            blob = bytes(_encode_view(xs_l, model_l, clog, 1)[0])
            te_l = timed2(lambda: _encode_view(xs_l, model_l, clog, 1), reps)
            td_l = timed2(lambda: _decode_bytes(blob, model_l, device), reps)
            out_l, _, st_l = _decode_bytes(blob, model_l, device)
            good = bool(np.array_equal(sorted_rows(out_l.cpu().numpy()), sorted_rows(pts_l)))
            return blob, te_l, td_l, st_l, good

        n_l = args.points
        blob0, te0, td0, st0_, ok0 = layout_point(x, pts, model, 0, 2)
        reference_layout = {"container": "v0: one torchac-compatible stream per level and stage (pcc_utils.py:174-177, :198-203), chunk_log2 = 0",
                            "points": n_l, "enc_ms": round(te0 * 1e3, 3), "dec_ms": round(td0 * 1e3, 3), "value": round(n_l / (te0 + td0) / 1e6, 4),
                            "unit": "Mpoints/s", "bytes": len(blob0), "bpp": round(len(blob0) * 8 / n_l, 4), "roundtrip_bit_identical": ok0,
                            "coder": "host side (csrc/hostcoder.hip: torchac's loop; encode: the independent streams on up to 16 native threads, decode: one thread -- "
                                     "stage s + 1 needs stage s's symbols); the network runs on the device",
                            # four symbols per coded node; a decode's streams are ONE dependent chain: the rate is what one host thread decodes
                            "coded_symbols": int(4 * st0_.coded_nodes),
                            "decode_Msymbols_per_s": round(4 * st0_.coded_nodes / td0 / 1e6, 2), "encode_Msymbols_per_s": round(4 * st0_.coded_nodes / te0 / 1e6, 2)}
        # The chunk size trades bytes for decode latency (a lane of 2^(chunk_log2 - 1) symbols is one dependent chain; chunks cost ~2.3 bytes each):
        # the same cloud and weights at other chunk sizes, so that the operating point of `value` (chunk_log2 = 11) can be judged against its neighbours
        chunk_sweep = []
        for cl in (9, 10, 11, 12, 13):
            blob_c, te_c, td_c, _, ok_c = layout_point(x, pts, model, cl, 4)
            chunk_sweep.append({"chunk_log2": cl, "enc_ms": round(te_c * 1e3, 3), "dec_ms": round(td_c * 1e3, 3), "value": round(n_l / (te_c + td_c) / 1e6, 4),
                                "bytes": len(blob_c), "overhead_frac_vs_v0": round((len(blob_c) - len(blob0)) / len(blob0), 5), "roundtrip_bit_identical": ok_c})
        low_rate = []
        for label, cloud_fn, sd_fn in (
                ("bench cloud, peaky model (head biases = log stage-symbol frequencies, head weights x 0.25)", lambda: (x, pts), lambda p_: peaky_state_dict(32, k)),
                ("solid cloud (every voxel inside 12 seeded balls), its own peaky model (conv gain 1)",
                 lambda: (lambda c_: (torch.tensor(c_, device=device), c_))(solid_cloud(n_l)), lambda p_: peaky_state_dict(32, k, gain=1.0, freq=stage_symbol_frequencies(p_)))):
            xs_l, pts_l = cloud_fn()
            model_l = runtime.Model(sd_fn(pts_l), 32, k, local_rank)
            blob4, te4, td4, st4, ok4 = layout_point(xs_l, pts_l, model_l, args.chunk_log2, 5)
            blobv0, tev0, tdv0, _, okv0 = layout_point(xs_l, pts_l, model_l, 0, 1)
            # the dominant kernel on THIS cloud (HIP events around the conv launches of one encode + decode, as `roofline` does for the headline):
            # the solid cloud's levels are dense (every node has most of its 125 neighbours), its tiles nearly full
            _lib.check(L.gpcc_profile_enable(ctx, 1))
            _decode_bytes(bytes(_encode_view(xs_l, model_l, args.chunk_log2, 1)[0]), model_l, device)
            torch.cuda.synchronize(device)
            prof_l = _lib.Profile()
            _lib.check(L.gpcc_profile_get(ctx, C.byref(prof_l)))
            _lib.check(L.gpcc_profile_enable(ctx, 0))
            conv_tf = 2.0 * 32 * 32 * prof_l.conv_pair_jobs / (prof_l.conv_ms * 1e-3) / 1e12 if prof_l.conv_ms > 0 else 0.0
            low_rate.append({"case": label, "points": n_l, "coded_nodes": int(st4.coded_nodes), "bpp": round(len(blob4) * 8 / n_l, 4),
                             "bits_per_coded_node": round(len(blob4) * 8 / max(1, st4.coded_nodes), 3),
                             "enc_ms": round(te4 * 1e3, 3), "dec_ms": round(td4 * 1e3, 3), "value": round(n_l / (te4 + td4) / 1e6, 4), "unit": "Mpoints/s",
                             "container_bytes": len(blob4), "bytes_v0": len(blobv0), "bpp_v0": round(len(blobv0) * 8 / n_l, 4),
                             "chunk_overhead_frac": round((len(blob4) - len(blobv0)) / len(blobv0), 5),
                             "v0_enc_ms": round(tev0 * 1e3, 3), "v0_dec_ms": round(tdv0 * 1e3, 3), "roundtrip_bit_identical": bool(ok4 and okv0),
                             "conv_TFLOP_per_s": round(conv_tf, 2), "conv_frac_of_mfma_peak": round(conv_tf / MFMA_F32_PEAK_TFLOPS, 4),
                             "conv_pairs_per_node": round(st4.conv_pairs / 18.0 / max(1, st4.coded_nodes), 1)})
            del model_l
        data, st = _encode_view(x, model, args.chunk_log2, 1)   # `data` is a view of the context's buffer: restore it

    # correctness of what was just timed: decoded geometry == input geometry (as sets; bit-identical)
    d = dec.cpu().numpy()
    ok = d.shape == pts.shape and np.array_equal(d[np.lexsort((d[:, 0], d[:, 1], d[:, 2]))], pts[np.lexsort((pts[:, 0], pts[:, 1], pts[:, 2]))])
    if not ok:
        raise SystemExit("round trip failed: decoded geometry differs from the input")

    # max over ranks; bitstream stats collated with one all_gather (RCCL over xGMI when N > 1)
    mine = SceneStats(num_points=args.points, num_bytes=len(data), enc_s=enc_s / args.steps, dec_s=dec_s / args.steps,
                      coded_nodes=st.coded_nodes, conv_pairs=st.conv_pairs, levels=st.num_levels, status=0)
    scenes = collate_stats([mine], device)
    elapsed = max_over_ranks(elapsed, device)
    allstats = np.array([[s.num_bytes, s.enc_s, s.dec_s, s.coded_nodes, s.conv_pairs] for s in scenes], dtype=np.float64)

    if rank == 0:
        # HBM traffic of the conv kernel comes from separate rocprofv3 --pmc passes of this same command
        # (tools/pmc_traffic.sh); the corrected per-launch figure is kept under profiles/
        traffic, traffic_source, pmc_extra = None, None, {}
        if not args.measure_traffic:
            for name in ("r06_pmc_conv.json", "r05_pmc_conv.json", "r04_pmc_conv.json", "r03_pmc_conv.json", "r02_pmc_conv.json"):
                try:
                    with open(os.path.join(ROOT, "profiles", name)) as f:
                        pj = json.load(f)
                    traffic = round(pj["hbm_bytes_per_launch"])
                    traffic_source = f"profiles/{name} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, tools/pmc_traffic.sh; not measured in this run)"
                    # the other factors of frac, from the SQ / GRBM counter passes of the same round (labelled like traffic: replayed, not live)
                    for key in ("mfma_busy", "tile_fill"):
                        if key in pj:
                            pmc_extra[key] = pj[key]
                    if pmc_extra:
                        pmc_extra["counters_source"] = f"profiles/{name} (tools/pmc_conv2.sh passes of this command; not measured in this run)"
                    break
                except Exception:
                    pass
        total_points = args.points * world * K * args.steps
        value = total_points / elapsed / 1e6
        conv_flops = 2.0 * 32 * 32 * prof.conv_pair_jobs
        achieved = conv_flops / (prof.conv_ms * 1e-3) / 1e12 if prof.conv_ms > 0 else 0.0
        out = {
            "metric": "Mpoints/s encode+decode @1M anchors",
            "value": round(value, 4),
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "GausPcgc encode+decode of one synthetic anchor cloud per GPU (BASELINE configs[1])",
                "points_per_scene": args.points,
                "scenes_per_gpu": K,
                "scenes_share_launches": bool(batch_mode),   # K > 1: gpcc_encode_batch / gpcc_decode_batch (csrc/forest.hpp)
                "channels": 32,
                "kernel_size": k,
                "container": f"v{data[2]} (per-level chunks of two coder lanes, chunk_log2<={args.chunk_log2}" + ("; carry-propagating range coder in the lanes)" if data[2] >= 4 else ")") if args.chunk_log2 else "v0 (reference layout)",
                "weights": "seeded synthetic (reference initialisers, conv gain 4)",
            },
            "enc_ms": round(float(allstats[:, 1].mean()) * 1e3 / K, 3),     # per scene
            "dec_ms": round(float(allstats[:, 2].mean()) * 1e3 / K, 3),
            # aggregate encode-only rate of the batch (configs[3] is an encode batch): all scenes / the slowest rank's encode time
            "enc_value": round(args.points * world * K / float(allstats[:, 1].max()) / 1e6, 4),
            "bpp": round(float(allstats[:, 0].mean()) * 8 / args.points, 4),
            # The chunked container pays a fixed number of bytes per chunk (~5 bits of chunk table + two coder flushes) for its parallel decode.
            # The seeded random weights code ~22 bpp; a trained model at a few bpp shrinks the payload, not this overhead, so
            # it is also quoted against a 4 bpp payload (HAC-class rates).  chunk_log2 = 0 writes bytes_v0 exactly.
            "container_bytes": len(data),
            "bytes_v0": bytes_v0,
            "bpp_v0": None if bytes_v0 is None else round(bytes_v0 * 8 / args.points, 4),
            "chunk_overhead_bytes": None if bytes_v0 is None else len(data) - bytes_v0,
            "chunk_overhead_frac": None if bytes_v0 is None else round((len(data) - bytes_v0) / bytes_v0, 5),
            "chunk_overhead_frac_at_4bpp": None if bytes_v0 is None else round((len(data) - bytes_v0) / (4.0 * args.points / 8), 5),
            "scenes_in_flight": inflight,
            "sizes": sizes,
            "batched": batched,
            "reference_layout": reference_layout,
            "low_rate": low_rate,
            "chunk_sweep": chunk_sweep,
            "coded_nodes": int(allstats[0, 3]),
            "ranks": [{"bytes": int(r_[0]), "coded_nodes": int(r_[3]), "enc_ms": round(float(r_[1]) * 1e3, 3), "dec_ms": round(float(r_[2]) * 1e3, 3)} for r_ in allstats],
            "roundtrip_bit_identical": True,
            "roofline": {
                "kernel": "k_sparse_conv",
                "bound": "mfma",
                "achieved": round(achieved, 3),
                "peak": MFMA_F32_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                **pmc_extra,
                "algorithmic_per_launch": conv_flops / max(prof.conv_launches, 1),
                "launches": int(prof.conv_launches),
                "avg_launch_us": round(prof.conv_ms * 1e3 / max(prof.conv_launches, 1), 2),
                "algorithmic_flops_per_step": conv_flops / max(ev_steps, 1),
                "conv_time_frac_of_step": round(prof.conv_ms * 1e-3 / max(ev_steps, 1) / ((enc_s + dec_s) / args.steps), 4),
                # every bracket is a hipEventRecord on the kernel's stream and costs it ~5 us; `event_steps` of the timed
                # steps carry them (launches / avg_launch_us / achieved are over those steps)
                "event_steps": ev_steps,
                # the HBM-bound stages either side of the convolutions: algorithmic bytes (HISTORY.md section 4) over the
                # event-bracketed time of one extra untimed step; latency- / launch-bound at this size, not bandwidth-bound
                "stages": stages,
                # the decoder's levels of at most 16 k nodes run as persistent launches (csrc/fused.hpp), not as k_sparse_conv: their time
                # covers whole chains of layers (heads, coder phases and grid barriers too), so they are reported apart from `achieved`
                "small_levels": {
                    "kernel": "k_level_fused",
                    "ms_per_step": round(prof.fused_ms / max(ev_steps, 1), 3),
                    "convolutions": int(prof.fused_launches),
                    "algorithmic_flops_per_step": 2.0 * 32 * 32 * prof.fused_pair_jobs / max(ev_steps, 1),
                },
                # the same fraction with those launches counted WHOLE as convolution time (comparable with the lines of rounds 1-3, where
                # `frac` covered every level; conservative: their heads, coder phases and barriers are in the denominator)
                "frac_all_levels": round((conv_flops + 2.0 * 32 * 32 * prof.fused_pair_jobs) / max((prof.conv_ms + prof.fused_ms) * 1e-3, 1e-9) / 1e12 / 157.3, 4),
            },
        }
        if args.side_anchors > 0 and world == 1:
            # the callers either side of the path, measured behind the timed region on a synthetic
            # scene: seconds / ms / symbol rates as observed in THIS run (tools/bench_side_paths.py holds the harness)
            import importlib.util

            spec = importlib.util.spec_from_file_location("bench_side_paths", os.path.join(ROOT, "tools", "bench_side_paths.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            import contextlib

            with contextlib.redirect_stdout(sys.stderr):   # the reference-compatible callers print progress lines (`Start encoding ...`,
                sp = mod.measure(args.side_anchors)        # HAC/scene/gaussian_model.py:1095): stdout carries the ONE JSON line only
            out["side_paths"] = {
                "scene": f"synthetic HAC-style scene, {sp['n_anchors']} anchors x 50 features x 10 offsets, {sp['image'][0]}x{sp['image'][1]} frame",
                "attribute_loop": {k: sp["attribute_loop"][k] for k in ("anchors_coded", "files_bytes", "conduct_encoding_s", "conduct_decoding_s")},
                "attribute_loop_hac_plus": {k: v for k, v in sp.get("attribute_loop_hac_plus", {}).items() if k != "log"},
                "gaussian_coder": {k: sp["gaussian_coder"][k] for k in ("symbols", "encode_fused_ms", "decode_fused_ms", "Msymbols_per_s_encode", "Msymbols_per_s_decode",
                                                                        "fused_bytes_equal_table_bytes")},
                "mlp_grid": sp["mlp_grid"],
                "rd_loop": {k: sp["rd_loop"][k] for k in ("gaussians", "generate_neural_gaussians_ms", "generate_plus_rasterise_ms", "psnr_decoded_vs_encoder_side_dB")},
                "torchac_shim": sp["torchac_shim"],
            }
        if args.cpu_sample > 0 and world == 1:   # the CPU baseline is a rank-0, N = 1 figure
            from gauspcc_amd.model import tensor_table
            from oracle import oracle as orc

            host_cores = len(os.sched_getaffinity(0))
            om = orc.Model(tensor_table(sd, 32, k), 32, k)
            calib = {}
            if args.cpu_threads > 0:
                threads = orc.set_threads(max(1, min(args.cpu_threads, host_cores)))
            else:
                # the baseline must not be handicapped by its thread count in either direction: every core of a 256-core box ran the
                # oracle 10x SLOWER than 64 threads (round 4, first try: 56.8 s against 5.7 s per 1 M-point encode), a fixed 64 may
                # be too few elsewhere -- so a short calibration (encode of a 100 k-point cloud) picks among a few counts
                small = synthetic_cloud(min(100_000, args.cpu_sample), seed=seed + 1)
                for cand in sorted({c for c in (8, 16, 32, 64, 128, host_cores) if c <= host_cores}):
                    orc.set_threads(cand)
                    c0 = time.perf_counter()
                    orc.encode(om, small, chunk_log2=args.chunk_log2)
                    calib[cand] = round(time.perf_counter() - c0, 3)
                    if calib[cand] > 4.0 * min(calib.values()):
                        break
                threads = orc.set_threads(min(calib, key=calib.get))
            sp = synthetic_cloud(args.cpu_sample, seed=seed)
            t0 = time.perf_counter()
            ref = orc.encode(om, sp, chunk_log2=args.chunk_log2)
            t1 = time.perf_counter()
            od, _ = orc.decode(om, ref)
            t2 = time.perf_counter()
            assert od.shape == sp.shape
            out["cpu_baseline"] = {
                "value": round(args.cpu_sample / (t2 - t0) / 1e6, 5),
                "unit": "Mpoints/s",
                "cores": threads,            # the threads actually used: the fastest of the calibrated counts unless --cpu-threads pins one
                "threads": threads,
                "host_cores": host_cores,
                "thread_calibration_s": calib or None,   # seconds per 100 k-point oracle encode at each candidate thread count
                "kind": "port",
                "sample": f"oracle encode+decode of a {args.cpu_sample}-point cloud from the same generator "
                          f"(enc {t1 - t0:.2f} s, dec {t2 - t1:.2f} s; {threads} OpenMP threads for the convolutions and heads, "
                          f"single-thread range coder as torchac; the box has {host_cores} cores; thread count {'pinned' if args.cpu_threads > 0 else 'picked by calibration'})",
            }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
