"""ConsensusGenerator::generateConsensuses over the golden pairs (tests/golden/consensus_pairs.json), by the reference
alone (1 and N threads) and by the program with the device seams; prints the times and what the seams did.
    python tools/consensus_bench.py [threads] [repeat]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import edit_pair
from oracle import oracle as O
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 1
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "consensus_pairs.json")))
pairs = [edit_pair(s) for s in gold["specs"]] * rep
os.environ["FLYE_GPU_STATS"] = "/tmp/consensus_seam.json"
for name, binary, th in (("device seams", O.REF_DUMPER_GPU, threads), ("reference", None, 1), ("reference", None, threads)):
    text, info = O.ref_consensus(pairs, threads=th, binary=binary)
    print(f"{name:13s} {th:3d} threads: {len(pairs)} pairs in {info['consensus_s'] * 1e3:8.1f} ms = {len(pairs) / info['consensus_s']:8.0f} pairs/s", flush=True)
    if binary and os.path.exists("/tmp/consensus_seam.json"):
        print("   seams:", open("/tmp/consensus_seam.json").read().strip())
