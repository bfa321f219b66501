"""Write a seeded synthetic dataset in the reference's binary format (SURVEY Appendix B /
include/gnn/reader.h) so the trainer CLI can be exercised at the BASELINE configs' sizes:

    python scripts/make_synth_dataset.py ogbn-products /tmp/data      # -> /tmp/data/ogbn-products/graph.*
    DATASET_PATH=/tmp/data/ ./bin/gpu_train_sage ogbn-products 10 32 softmax 256 0 0 0.01 3 0 50 0

Topology: graphaibench_amd.synth (Chung-Lu, shaped like the named dataset); features: class-mean +
noise so that training has something to learn; labels uniform; masks = contiguous 8 % / 2 % / 90 %.
"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from graphaibench_amd import synth  # noqa: E402


def cora(root: Path, feat_len: int = 1433, seed: int = 0):
    """the reference's own cora topology with seeded features (synth.write_cora_dataset)"""
    info = synth.write_cora_dataset(root, Path(__file__).resolve().parent.parent / "tests" / "golden" / "cora", feat_len, seed)
    print(f"wrote {info['dir']}: nv={info['nv']} F={info['F']} C={info['C']}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name", choices=list(synth.SHAPES) + ["cora"])
    ap.add_argument("root")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    args = ap.parse_args()
    if args.name == "cora":
        return cora(Path(args.root))
    info = synth.write_dataset(args.name, args.root, scale=args.scale, device=args.device)
    print(f"wrote {info['dir']}: nv={info['nv']} ne={info['ne']} F={info['F']} C={info['C']} max_degree={info['max_degree']}")


if __name__ == "__main__":
    main()
