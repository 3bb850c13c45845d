"""Which library calls consume values of libc's rand() stream?  (Bit parity of
OverlapContainer::estimateOverlaperParameters, overlap.cpp:752-756, needs the stream untouched.)
After each phase: srand(1) was called before it; the next rand() must be 1804289383."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flye_amd import config, gpu, synth
libc = ctypes.CDLL(None)
FIRST = 1804289383
def phase(name, fn):
    libc.srand(1)
    r = fn()
    v = libc.rand()
    print(f"{name:34s} next rand() = {v}  {'untouched' if v == FIRST else 'CONSUMED'}", flush=True)
    return r
rs = synth.simulate(seed=7, genome_len=40_000, coverage=20, kind="hifi").filter_min_len(1000)
cfg = config.preset("hifi")
ctx = phase("fg_create (HIP init)", lambda: gpu.Context(17, 0))
phase("fg_set_reads", lambda: ctx.set_reads(rs))
vi = gpu.VertexIndex(ctx, 2.0)
phase("fg_build_index_minimizers", lambda: vi.build(cfg))
det = gpu.OverlapDetector.for_assemble(ctx, vi, cfg)
q = np.arange(0, 2 * rs.n, 2, dtype=np.uint32)
phase("first fg_overlaps (kernel loading)", lambda: det.getSeqOverlapsBatch(q))
phase("second fg_overlaps", lambda: det.getSeqOverlapsBatch(q))
ctx2 = phase("second fg_create", lambda: gpu.Context(17, 0))
ctx2.set_reads(rs)
vi2 = gpu.VertexIndex(ctx2, 1.0)
phase("fg_build_index_solid", lambda: (vi2.countKmers(), vi2.buildIndexUnevenCoverage(2, 0.4, 100)))
