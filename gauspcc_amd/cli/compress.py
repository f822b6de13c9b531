"""Compress point cloud geometry files (reference CLI: src/ai_pcc/GausPcgc/compress_ue_4stage_conv.py:32-288).

Same flags, defaults, `.bin` naming (<file name>.bin in --output_folder), CSV
(<resultdir>/<prefix>_data<N>.csv with columns filedir,bpp,enc_time,file_size_bits,num_points and a
final `avg` row) and summary line.  The per-file codec is gauspcc_amd.pcc_utils.compress_point_cloud
(libgauspcc on the MI355X).  --chunk_log2 0 writes the reference's container.
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
    p.add_argument("--chunk_log2", type=int, default=None, help="extension: 0 = reference container, 6..14 = chunked v1 container (default 10)")
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


def main(argv=None):
    args = build_parser().parse_args(argv)
    from . import io

    io.export_hw_queues(args.jobs)
    import torch

    from .. import pcc_utils, runtime

    os.makedirs(args.output_folder, exist_ok=True)
    os.makedirs(args.resultdir, exist_ok=True)
    files = list_inputs(args.input_glob, args.num_samples)
    if not files:
        raise SystemExit(f"no input files match {args.input_glob}")
    xyz_ls = io.read_point_clouds(files)
    device = torch.device("cuda", torch.cuda.current_device())
    runtime.get_model(args.ckpt, args.channels, args.kernel_size, device)   # load once, outside the timed span (:67-70)
    # warm-up like :72-75: one small random cloud through the whole path
    warm = torch.unique(torch.randint(0, 2048, (2048, 3), dtype=torch.int32), dim=0).to(device)
    pcc_utils._encode_to_bytes(warm, runtime.get_model(args.ckpt, args.channels, args.kernel_size, device), 0, 1)

    rows = []
    csvfile = os.path.join(args.resultdir, args.prefix + "_data" + str(len(files)) + ".csv")

    def one(item):
        path, pts = item
        name = os.path.split(path)[-1]
        xyz = quantise(pts, args.is_data_pre_quantized, args.posQ, device)
        n_in = len(pts)
        r = pcc_utils.compress_point_cloud(xyz, args.ckpt, os.path.join(args.output_folder, name + ".bin"), channels=args.channels,
                                           kernel_size=args.kernel_size, posQ=args.posQ, chunk_log2=args.chunk_log2)
        return {"filedir": name, "bpp": r["file_size_bits"] / n_in, "enc_time": r["enc_time"], "file_size_bits": r["file_size_bits"], "num_points": n_in}

    if args.jobs <= 1:
        for item in zip(files, xyz_ls):
            rows.append(one(item))
            write_results_csv(rows, csvfile, with_avg=False)   # the reference rewrites the CSV after every file
    else:
        rows = io.run_jobs(one, zip(files, xyz_ls), args.jobs)
    write_results_csv(rows, csvfile, with_avg=True)
    print("Total: {total_n:d} | Average bitrate:{bpp:.3f} | Encoding time:{enc_time:.3f}s | Max GPU memory:{memory:.2f}MB".format(
        total_n=len(rows), bpp=np.mean([r["bpp"] for r in rows]), enc_time=np.mean([r["enc_time"] for r in rows]),
        memory=torch.cuda.max_memory_allocated() / 1024 / 1024))
    print("Results saved to ", csvfile)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
