"""BASELINE configs[4] (CHM13 HiFi 30x on 8 GPUs) at a chosen fraction of its size, rank 0's part on ONE GPU: the same run
as tests/test_gpu_configs.py::test_config5_hifi_proxy_rank0_of_8_bounded_memory (one-GPU yardstick index, rank-0 build
within a stated memory bound, in-place gather, rank 0's queries, sampled oracle parity) with the genome length given.
    python tools/config5_scale.py 1000000000      # 1 Gb genome, 30 Gbp of reads: a third of CHM13"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_configs as T
T.config5_rank0_of_8(int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000, n_sample=int(sys.argv[2]) if len(sys.argv) > 2 else 120)
