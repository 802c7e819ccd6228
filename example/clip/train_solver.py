"""Entry point with the reference's CLI (example/clip_fdt/train_solver.py:723-747, example/clip/train_solver.py:629-655):

    torchrun --nproc_per_node 8 --master-addr 127.0.0.1 example/clip_fdt/train_solver.py \
        --config example/clip_fdt/config_cc3m.yaml --output_path out --batch_size 256 [--synthetic]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ilvlm_amd.solver import main  # noqa: E402

if __name__ == "__main__":
    main()
