import sys; sys.path.insert(0, "/root/repo")
import time
from flye_amd import config, gpu, workloads
rs, mo, preset = workloads.ecoli_pb50()
cfg = config.preset(preset)
ctx = gpu.Context(17, 0); ctx.set_reads(rs)
vi = gpu.VertexIndex(ctx, 1.0)
for rep in range(3):
    t = time.time(); st = vi.build(cfg); dt = time.time() - t
    kt = ctx.kernel_times()
    print("build wall %.1f ms, reported %.1f ms" % (dt * 1e3, st["build_seconds"] * 1e3))
    print({k: round(v[0] * 1e3, 2) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])[:14]}, "sum %.1f" % (sum(v[0] for v in kt.values()) * 1e3))
