"""Times sharded.echelonize_sharded against the single-device echelonize (ranks share cuda:0, collectives over gloo: a
rehearsal of the protocol, not a multi-GPU measurement):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 tools/time_sharded.py [n=100000] [row_nnz=6]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import spasm_jl_amd as S
from spasm_jl_amd import sharded
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dist.init_process_group("gloo")
torch.cuda.set_device(0)
A = S.synth_csr(1, n, n, row_nnz=k, prime=65521, seed=21)
dist.barrier(); t = time.time()
fact, info = sharded.echelonize_sharded(A, finish_nnz=int(os.environ.get("FINISH_NNZ", 50000)))
dist.barrier(); dt = time.time() - t
if dist.get_rank() == 0:
    print(f"sharded x{dist.get_world_size()}: {dt:.2f}s rank {fact.r} rounds {[(r['finish'], r['npiv']) for r in info['rounds']]}", flush=True)
    for r in info["rounds"]:
        print("  round", r["round"], "rows", r["rows"], "nnz", r["nnz"], {k: round(v, 3) for k, v in r["seconds"].items()}, flush=True)
    t = time.time(); ref = S.echelonize(A); dt1 = time.time() - t
    print(f"single device: {dt1:.2f}s rank {ref.r} rounds {[r['npiv'] for r in S.last_rounds()]}", flush=True)
    assert ref.r == fact.r and S.factorization_verify(A, fact, 1)
dist.destroy_process_group()
