"""Multi-rank EM on the device halves of the C ABI (rsem_em_estep_device / rsem_em_mstep_device): two ranks share
GPU 0, reads sharded by the reference's rule, counts all-reduced each round (gloo here; RCCL in bench.py).
The result must equal a single-context run on the whole matrix."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from rsem_amd import capi, dist as rd
    from tools.synth_data import make_em_workload
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    wl = make_em_workload("small", seed=9)
    M = wl["M"]
    b = rd.shard_rows(wl["row_ptr"], world)
    rp, sid, cp, ncp = rd.take_shard(wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], b[rank], b[rank + 1])
    ctx = capi.EmContext(M, rp, np.ascontiguousarray(sid), np.ascontiguousarray(cp), np.ascontiguousarray(ncp), device=0)
    theta = [torch.from_numpy(wl["theta0"]).to(dev), torch.zeros(M + 1, dtype=torch.float64, device=dev)]
    counts = torch.zeros(M + 1, dtype=torch.float64, device=dev)
    stats = torch.zeros(3, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for r in range(6):
        a, bb = theta[r & 1], theta[(r + 1) & 1]
        ctx.estep_device(a.data_ptr(), counts.data_ptr(), stream)
        dist.all_reduce(counts)
        ctx.mstep_device(counts.data_ptr(), float(wl["N0"]), a.data_ptr(), bb.data_ptr(), stats.data_ptr(), stream)
    torch.cuda.synchronize()
    if rank == 0:
        q.put((theta[0].cpu().numpy(), stats.cpu().numpy()))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


def test_two_rank_em_equals_single_context():
    import torch.multiprocessing as mp
    from rsem_amd import capi
    from tools.synth_data import make_em_workload
    ctx_mp = mp.get_context("spawn")
    q = ctx_mp.Queue()
    port = 29700 + os.getpid() % 1000
    procs = [ctx_mp.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    theta_d, stats = q.get(timeout=300)
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    wl = make_em_workload("small", seed=9)
    ctx = capi.EmContext(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
    out = ctx.run(wl["theta0"], wl["N0"], min_round=6, max_round=6)
    assert np.allclose(theta_d, out["theta"], rtol=1e-9, atol=1e-18)
    assert abs(stats[0] - (wl["N0"] + len(wl["row_ptr"]) - 1)) < 1e-6
    assert int(stats[2]) == out["totNum"]
    ctx.close()
