import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["PLO_LANE_HEAVY_MIN"] = "0"; os.environ["PLO_LANE_MAX_W"] = "12"; os.environ["PLO_LANE_HEAVY_PER"] = sys.argv[1] if len(sys.argv) > 1 else "5"
import fuzz_cases
from oracle import pyoracle
from portello_amd import abi, api
pyoracle.build()
import torch
def dirty():
    xs = [torch.randint(-2**31, 2**31 - 1, (64 << 20,), dtype=torch.int32, device='cuda') for _ in range(8)]
    torch.cuda.synchronize(); del xs; torch.cuda.empty_cache()
for seed in range(40):
    alpha = (b"ACGT", b"AC", b"A")[seed % 3]
    ix, b = fuzz_cases.make(1000 + seed, alphabet=alpha, n_reads=120, explicit=(seed % 3 == 0), seq_fmt=(abi.SEQ_BAM4 if seed % 4 == 1 else abi.SEQ_ASCII))
    dirty()
    index = api.Index(ix); eng = api.Engine(index)
    for stages in (31, 15, 5, 2, 16, 7, 27):
        print("seed", seed, "stages", stages, flush=True)
        got = eng.liftover_batch(b, stages)
        t = eng.timing()
        ok = pyoracle.liftover_batch(ix, b, stages, 1).canonical() == got.canonical()
        print("   ok" if ok else "   MISMATCH", t.n_heavy_lane_items, t.n_retry_items, t.n_big_items, flush=True)
    eng.close(); index.close()
