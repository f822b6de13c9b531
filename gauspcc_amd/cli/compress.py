"""Compress point cloud geometry files (reference CLI: src/ai_pcc/GausPcgc/compress_ue_4stage_conv.py:32-288).

Same flags, defaults, `.bin` naming (<file name>.bin in --output_folder), CSV
(<resultdir>/<prefix>_data<N>.csv with columns filedir,bpp,enc_time,file_size_bits,num_points and a
final `avg` row) and summary line.  The per-file codec is gauspcc_amd.pcc_utils.compress_point_cloud
(libgauspcc on the MI355X).  --chunk_log2 0 writes the reference's container.

Batches shard over the GPUs of a node (BASELINE configs[3]): `--gpus N`, or the same command under
`python -m torch.distributed.run --nproc-per-node N --module gauspcc_amd.cli.compress ...`.  File i of the sorted list
goes to rank i mod N; every rank has its own device, model replica and output files, nothing is exchanged on the data
path; one all_gather collates the per-file rows (gauspcc_amd.dist.collate_stats) and rank 0 writes the ONE CSV, in
input order with the `avg` row, exactly as a single process would.
"""
import argparse
import os
from glob import glob

import numpy as np


def build_parser():
    p = argparse.ArgumentParser(prog="gauspcc_amd.cli.compress", description="Compress point cloud geometry data using unequal 4-stage convolution network",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--input_glob", default="./data/kittidet_examples/*.ply", help="Glob pattern for input point cloud files")
    p.add_argument("--output_folder", default="./data/kittidet_compressed/", help="Folder to save compressed bin files")
    p.add_argument("--is_data_pre_quantized", type=bool, default=False, help="Whether input data is pre-quantized")
    p.add_argument("--posQ", default=16, type=int, help="Quantization scale")
    p.add_argument("--channels", type=int, help="Neural network channel count", default=32)
    p.add_argument("--kernel_size", type=int, help="Convolution kernel size", default=3)
    p.add_argument("--ckpt", help="Checkpoint loading path ('synthetic[:seed]' = seeded random weights)", default="./model/KITTIDetection/ckpt_ue_4stage_conv.pt")
    p.add_argument("--num_samples", default=-1, type=int, help="Use the first N files for quick testing. [-1 means test all data]")
    p.add_argument("--resultdir", type=str, default="./results", help="Folder to save result CSV files")
    p.add_argument("--prefix", type=str, default="ue_4stage_conv", help="Prefix for result CSV files")
    p.add_argument("--jobs", type=int, default=1, help="extension: files in flight on the GPU (host threads with their own stream and context: one file's host work and idle device time overlap another's kernels)")
    p.add_argument("--chunk_log2", type=int, default=None, help="extension: 0 = reference container, 6..14 = chunked container (default: pcc_utils.DEFAULT_CHUNK_LOG2)")
    p.add_argument("--gpus", type=int, default=1, help="extension: shard the files over this many GPUs of the node, one process per GPU (file i -> rank i mod N)")
    p.add_argument("--batch", type=int, default=1, help="extension: code this many files of a rank's share through ONE chain of launches (pcc_utils.compress_point_clouds; "
                   "small clouds leave most of the GPU idle one at a time).  Same .bin files; a file's enc_time is its batch's span divided by the batch size")
    p.add_argument("--selftest-stub", action="store_true", help=argparse.SUPPRESS)   # tests/test_dist_cpu.py: the sharding / collation path without a GPU
    return p


def list_inputs(input_glob, num_samples=-1):
    """Reference :56-62 globs <input_glob>/**/*.* (the flag is used as a directory); a plain glob pattern is honoured too."""
    files = sorted(glob(os.path.join(input_glob, "**", "*.*"), recursive=True)) or sorted(glob(input_glob, recursive=True))
    files = [f for f in files if f.endswith(("h5", "ply", "bin", "npy"))]
    return files[:num_samples] if num_samples > 0 else files


def quantise(xyz, is_data_pre_quantized, posQ, device=None):
    """:89-94 on the device (gpcc_voxelise), in the dtype the reader produced (float32 for KITTI .bin, float64 for the text
    and PLY readers) as the reference's numpy / torch expressions do; coincident voxels merge, as the reference's sparse
    tensor construction merges them.  Returns an (M,3) int32 tensor on the device."""
    import torch

    from .. import pcc_utils

    a = np.asarray(xyz)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    device = device or torch.device("cuda", torch.cuda.current_device())
    q = pcc_utils.voxelise(torch.from_numpy(np.ascontiguousarray(a)).to(device), bool(is_data_pre_quantized), posQ)
    return torch.unique(q, dim=0)


def write_results_csv(rows, csvfile, with_avg):
    import pandas as pd

    df = pd.DataFrame(rows, columns=["filedir", "bpp", "enc_time", "file_size_bits", "num_points"])
    if with_avg:
        avg = df.mean(numeric_only=True).to_dict()
        avg["filedir"] = "avg"
        df = pd.concat([df, pd.DataFrame([avg])], ignore_index=True)
    df.to_csv(csvfile, index=False)
    return df


def _stub_codec(path, pts, out_path):
    """Deterministic stand-in for the device codec (tests of the sharding path on CPU ranks): a 'bitstream' whose size
    depends on the file's content only."""
    import zlib

    a = np.ascontiguousarray(np.asarray(pts, dtype=np.float64))
    blob = zlib.compress(a.tobytes(), 1)
    with open(out_path, "wb") as f:
        f.write(blob)
    return {"file_size_bits": 8 * len(blob), "enc_time": 0.001 * (1 + len(a) % 7)}


def main(argv=None):
    args = build_parser().parse_args(argv)
    from . import io

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # start the ranks ourselves, BEFORE anything here touches a GPU (gauspcc_amd.dist.launch_ranks); relay their exit code
        import sys

        from ..dist import launch_ranks

        return launch_ranks("gauspcc_amd.cli.compress", sys.argv[1:] if argv is None else list(argv), args.gpus, module=True)
    io.export_hw_queues(args.jobs)
    import torch

    from .. import dist as gdist

    rank, world, device = gdist.init_from_env(prefer_gpu=not args.selftest_stub)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    os.makedirs(args.output_folder, exist_ok=True)
    os.makedirs(args.resultdir, exist_ok=True)
    files = list_inputs(args.input_glob, args.num_samples)
    if not files:
        raise SystemExit(f"no input files match {args.input_glob}")
    mine = gdist.scenes_for_rank(len(files), rank, world)          # file i -> rank i mod world
    my_files = [files[i] for i in mine]
    xyz_ls = io.read_point_clouds(my_files) if my_files else []
    if not args.selftest_stub:
        from .. import pcc_utils, runtime

        runtime.get_model(args.ckpt, args.channels, args.kernel_size, device)   # load once, outside the timed span (:67-70)
        # warm-up like :72-75: one small random cloud through the whole path
        warm = torch.unique(torch.randint(0, 2048, (2048, 3), dtype=torch.int32), dim=0).to(device)
        pcc_utils._encode_to_bytes(warm, runtime.get_model(args.ckpt, args.channels, args.kernel_size, device), 0, 1)

    rows = []
    csvfile = os.path.join(args.resultdir, args.prefix + "_data" + str(len(files)) + ".csv")

    def one(item):
        path, pts = item
        name = os.path.split(path)[-1]
        n_in = len(pts)
        out_path = os.path.join(args.output_folder, name + ".bin")
        if args.selftest_stub:
            r = _stub_codec(path, pts, out_path)
        else:
            xyz = quantise(pts, args.is_data_pre_quantized, args.posQ, device)
            r = pcc_utils.compress_point_cloud(xyz, args.ckpt, out_path, channels=args.channels,
                                               kernel_size=args.kernel_size, posQ=args.posQ, chunk_log2=args.chunk_log2)
            io.note_workspace(device)
        return {"filedir": name, "bpp": r["file_size_bits"] / n_in, "enc_time": r["enc_time"], "file_size_bits": r["file_size_bits"], "num_points": n_in}

    def group(items):
        """--batch K: K files through one chain of launches (the reference's batch column, pcc_utils.py:73); same rows as `one`."""
        if args.selftest_stub or len(items) == 1:
            return [one(it) for it in items]
        names = [os.path.split(p)[-1] for p, _ in items]
        xyzs = [quantise(pts, args.is_data_pre_quantized, args.posQ, device) for _, pts in items]
        res = pcc_utils.compress_point_clouds(xyzs, args.ckpt, [os.path.join(args.output_folder, nm + ".bin") for nm in names], channels=args.channels,
                                              kernel_size=args.kernel_size, posQ=args.posQ, chunk_log2=args.chunk_log2)
        io.note_workspace(device)
        return [{"filedir": nm, "bpp": r["file_size_bits"] / len(pts), "enc_time": r["enc_time"], "file_size_bits": r["file_size_bits"], "num_points": len(pts)}
                for nm, (_, pts), r in zip(names, items, res)]

    if args.batch > 1:
        items = list(zip(my_files, xyz_ls))
        for g0 in range(0, len(items), args.batch):
            rows.extend(group(items[g0:g0 + args.batch]))
            if world == 1:
                write_results_csv(rows, csvfile, with_avg=False)
    elif args.jobs <= 1:
        for item in zip(my_files, xyz_ls):
            rows.append(one(item))
            if world == 1:
                write_results_csv(rows, csvfile, with_avg=False)   # the reference rewrites the CSV after every file
    else:
        # files in flight share the GPU: a file's enc_time is its wall-clock span UNDER CONTENTION (longer than alone; the
        # batch finishes sooner).  The CSV grows as files complete, in input order, so a failing job loses only its own row.
        # Under --gpus N every rank keeps --jobs files of ITS share in flight on its own GPU; only rank 0's single-rank runs
        # write the CSV incrementally (with several ranks the rows are collated at the end).
        def done(results):
            if world == 1:
                write_results_csv([r for r in results if r is not None], csvfile, with_avg=False)

        rows = io.run_jobs(one, zip(my_files, xyz_ls), args.jobs, on_progress=done)
    if world > 1:
        # one all_gather of the per-file records; every rank learns every row, rank 0 writes them in input order
        local = [gdist.SceneStats(num_points=r["num_points"], num_bytes=r["file_size_bits"] // 8, enc_s=r["enc_time"]) for r in rows]
        allst = gdist.collate_stats(local, device)
        order = gdist.scene_order(len(files), world)
        rows = [None] * len(files)
        for pos, s in zip(order, allst):
            rows[pos] = {"filedir": os.path.split(files[pos])[-1], "bpp": 8.0 * s.num_bytes / s.num_points, "enc_time": s.enc_s,
                         "file_size_bits": 8 * s.num_bytes, "num_points": s.num_points}
    if rank == 0:
        write_results_csv(rows, csvfile, with_avg=True)
        # torch's figure (the reference's column) cannot see the library's own workspace: add its high-water mark (this rank's contexts)
        mem = (torch.cuda.max_memory_allocated() + io.workspace_peak()) / 1024 / 1024 if device.type == "cuda" else 0.0
        print("Total: {total_n:d} | Average bitrate:{bpp:.3f} | Encoding time:{enc_time:.3f}s | Max GPU memory:{memory:.2f}MB".format(
            total_n=len(rows), bpp=np.mean([r["bpp"] for r in rows]), enc_time=np.mean([r["enc_time"] for r in rows]), memory=mem)
            + (f" | {world} ranks" if world > 1 else "") + (f" | {args.jobs} files in flight" + (" per rank" if world > 1 else "") + " (times under contention)" if args.jobs > 1 and args.batch <= 1 else "")
            + (f" | batches of {args.batch} (a file's time = its batch's span / the batch size)" if args.batch > 1 else ""))
        print("Results saved to ", csvfile)
    if world > 1:
        import torch.distributed as td

        td.barrier()
        td.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
