"""Decompress `.bin` files back to ASCII PLY (reference CLI: src/ai_pcc/GausPcgc/decompress_ue_4stage_conv.py:31-192).

Same flags and defaults; each <name>.bin becomes <output_folder>/<name>.bin.ply (:64) and the summary
line reports the mean decoding time.  The per-file codec is pcc_utils.decompress_point_cloud.
`--gpus N` (or torch.distributed.run --module gauspcc_amd.cli.decompress) shards the files like the compressor does:
file i -> rank i mod N, one all_gather of the per-file decoding times, rank 0 prints the one summary line.
"""
import argparse
import os
from glob import glob

import numpy as np


def build_parser():
    p = argparse.ArgumentParser(prog="gauspcc_amd.cli.decompress", description="Decompress point cloud geometry data using unequal 4-stage convolution network",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--input_glob", default="./data/kittidet_compressed/*.bin", help="Glob pattern for compressed bin files")
    p.add_argument("--output_folder", default="./data/kittidet_decompressed/", help="Folder to save decompressed ply files")
    p.add_argument("--is_data_pre_quantized", type=bool, default=False, help="Whether the original data was pre-quantized")
    p.add_argument("--channels", type=int, help="Neural network channel count", default=32)
    p.add_argument("--kernel_size", type=int, help="Convolution kernel size", default=3)
    p.add_argument("--ckpt", help="Checkpoint loading path ('synthetic[:seed]' = seeded random weights)", default="./model/KITTIDetection/ckpt_ue_4stage_conv.pt")
    p.add_argument("--jobs", type=int, default=1, help="extension: files in flight on the GPU (host threads with their own stream and context)")
    p.add_argument("--gpus", type=int, default=1, help="extension: shard the files over this many GPUs of the node, one process per GPU (file i -> rank i mod N)")
    p.add_argument("--batch", type=int, default=1, help="extension: decode this many files of a rank's share through ONE chain of launches (pcc_utils.decompress_point_clouds)")
    p.add_argument("--selftest-stub", action="store_true", help=argparse.SUPPRESS)   # tests/test_dist_cpu.py: the sharding / collation path without a GPU
    return p


def _stub_codec(path, out_path):
    """Inverse of compress._stub_codec (tests of the sharding path on CPU ranks): the zlib blob back to an ASCII PLY."""
    import zlib

    from ..pcc_utils import save_ply_ascii_geo

    with open(path, "rb") as f:
        a = np.frombuffer(zlib.decompress(f.read()), dtype=np.float64).reshape(-1, 3)
    save_ply_ascii_geo(a, out_path)
    return {"num_points": len(a), "dec_time": 0.001 * (1 + len(a) % 5)}


def main(argv=None):
    args = build_parser().parse_args(argv)
    from . import io

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import sys

        from ..dist import launch_ranks

        return launch_ranks("gauspcc_amd.cli.decompress", sys.argv[1:] if argv is None else list(argv), args.gpus, module=True)
    io.export_hw_queues(args.jobs)
    import torch

    from .. import dist as gdist

    rank, world, device = gdist.init_from_env(prefer_gpu=not args.selftest_stub)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    os.makedirs(args.output_folder, exist_ok=True)
    all_files = sorted(glob(args.input_glob))
    if not all_files:
        raise SystemExit(f"no input files match {args.input_glob}")
    files = [all_files[i] for i in gdist.scenes_for_rank(len(all_files), rank, world)]
    def one(path):
        name = os.path.split(path)[-1]
        if args.selftest_stub:
            r = _stub_codec(path, os.path.join(args.output_folder, name + ".ply"))
        else:
            from .. import pcc_utils

            r = pcc_utils.decompress_point_cloud(path, args.ckpt, os.path.join(args.output_folder, name + ".ply"), channels=args.channels,
                                                 kernel_size=args.kernel_size, is_data_pre_quantized=args.is_data_pre_quantized)
            io.note_workspace(device)
        print(f"Points after decompression: {r['num_points']}")
        return r["dec_time"]

    if args.batch > 1 and not args.selftest_stub:
        from .. import pcc_utils

        dec_time_ls = []
        for g0 in range(0, len(files), args.batch):
            grp = files[g0:g0 + args.batch]
            res = pcc_utils.decompress_point_clouds(grp, args.ckpt, [os.path.join(args.output_folder, os.path.split(p)[-1] + ".ply") for p in grp], channels=args.channels,
                                                    kernel_size=args.kernel_size, is_data_pre_quantized=args.is_data_pre_quantized)
            io.note_workspace(device)
            for r in res:
                print(f"Points after decompression: {r['num_points']}")
                dec_time_ls.append(r["dec_time"])
    else:
        dec_time_ls = io.run_jobs(one, files, args.jobs)
    if world > 1:
        allst = gdist.collate_stats([gdist.SceneStats(dec_s=t) for t in dec_time_ls], device)
        dec_time_ls = [s.dec_s for s in allst]
    if rank == 0:
        # torch's figure (the reference's column) cannot see the library's own workspace: add its high-water mark (this rank's contexts)
        mem = (torch.cuda.max_memory_allocated() + io.workspace_peak()) / 1024 / 1024 if device.type == "cuda" else 0.0
        print("Total: {total_n:d} | Decoding time:{dec_time:.3f}s | Max GPU memory:{memory:.2f}MB".format(
            total_n=len(dec_time_ls), dec_time=np.array(dec_time_ls).mean(), memory=mem)
            + (f" | {world} ranks" if world > 1 else "") + (f" | {args.jobs} files in flight" + (" per rank" if world > 1 else "") if args.jobs > 1 else ""))
    if world > 1:
        import torch.distributed as td

        td.barrier()
        td.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
