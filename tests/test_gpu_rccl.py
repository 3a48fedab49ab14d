"""RCCL executes on gfx950: a ONE-rank "nccl" process group on the one GPU of the test box, with the world-size-1 early-outs
of the exchange lifted (multiview.force_collectives), runs every collective shape the multi-GPU step issues -- async
all_reduce per piece, in-place reduce_scatter_tensor into a slice of its own input + in-place all_gather_into_tensor, the
packed row-sparse all_reduce, the MIN / MAX agreement all-reduces, all_gather_object + broadcast of the consistency term,
the statistics broadcasts, the parameter all-gather of ShardedFusedAdam -- through librccl's communicator and kernels on
the library's stream.  At world size 1 every one of them is a copy onto itself, so the results must equal the path that
issues no collective at all, bit for bit.  (The reference has no counterpart: it is single-process, train.py:171-240;
SURVEY.md section 5 "Distributed communication backend".)"""
import os

import pytest

from util import run_ranks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, types, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from splatco_amd import multiview
from splatco_amd.adam import FusedAdam, ShardedFusedAdam
from splatco_amd.densify import AnchorDensifier
from splatco_amd.multiview import GradArena
from splatco_amd.synthetic import synthetic_anchor_model, synthetic_views
from splatco_amd.train_step import collaborative_step

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
assert dist.get_backend() == "nccl"
print("RCCL", torch.cuda.nccl.version(), "on", torch.cuda.get_device_name(0), flush=True)
pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False)
bg = torch.ones(3, device=dev)
W, H, N, MV = 640, 360, 200_000, 2
views = [v.to(dev) for v in synthetic_views(MV, W, H)]
g = torch.Generator(device=dev).manual_seed(5)
base = torch.rand(3, H, W, device=dev, generator=g)
gts = [(base + 0.02 * i).clamp(0, 1) for i in range(MV)]


def make(force, mode, sparse=False, sharded=False):
    multiview.force_collectives(force)
    pc = synthetic_anchor_model(N, 9, dev, plane_size=256)
    idle = {id(p) for p in pc.feat_planes._feat.inactive_parameters()}
    groups = [{"params": [getattr(pc, "_" + n)], "lr": 1e-4, "name": n} for n in ("anchor", "offset", "anchor_feat", "scaling")]
    rest = [p for n, p in pc.named_parameters() if not n.startswith("_") and p.requires_grad and id(p) not in idle]
    groups.append({"params": rest, "lr": 1e-3, "name": "mlp_and_feat_planes"})
    params = [p for grp in groups for p in grp["params"]]
    arena = GradArena(params, chunk_bytes=4 << 20, mode=mode, anchor_ranges=4, sparse_rows=sparse, sparse_threshold=1.01, check_rows=True)
    assert arena.active == force and arena.world == 1
    opt = ShardedFusedAdam(groups, arena, eps=1e-15) if sharded else FusedAdam(groups, eps=1e-15)
    den = AnchorDensifier(pc, torch.optim.Adam(groups, eps=1e-15), voxel_size=0.01, seed=77)
    return pc, params, arena, opt, den


issued = {}
for name in ("all_reduce", "reduce_scatter_tensor", "all_gather_into_tensor", "broadcast", "all_gather_object"):
    def wrap(fn, name=name):
        def f(*a, **k):
            issued[name] = issued.get(name, 0) + 1
            return fn(*a, **k)
        return f
    setattr(dist, name, wrap(getattr(dist, name)))

for mode, cw, sparse, sharded in (("all_reduce", 0.05, False, False), ("rs_ag", 0.0, True, False), ("rs_ag", 0.0, False, True)):
    issued.clear()
    pc_a, params_a, arena_a, opt_a, den_a = make(True, mode, sparse, sharded)
    n_forced = dict(issued)
    pc_b, params_b, arena_b, opt_b, den_b = make(False, mode if not sharded else "all_reduce", False, False)
    for it in range(3):      # step 0 goes out from reduce(), the later ones from the hooks / ranges in the agreed order
        multiview.force_collectives(True)
        loss_a, _, _ = collaborative_step(pc_a, views, gts, pipe, bg, optimizer=opt_a, consistency_weight=cw, densifier=den_a,
                                          arena=arena_a, iteration=4 * (it + 1), tv_weight=1e-3)
        ga = [p.grad.clone() for p in params_a]
        forced = dict(issued)
        multiview.force_collectives(False)
        loss_b, _, _ = collaborative_step(pc_b, views, gts, pipe, bg, optimizer=opt_b, consistency_weight=cw, densifier=den_b,
                                          arena=arena_b, iteration=4 * (it + 1), tv_weight=1e-3)
        assert issued == forced, "the path without collectives issued one"
        assert float(loss_a) == float(loss_b), (mode, it, float(loss_a), float(loss_b))
        for i, (a, b) in enumerate(zip(params_a, params_b)):
            assert torch.equal(a, b), (mode, sharded, it, "parameter", i, tuple(a.shape))
        if not sharded:      # (the sharded step leaves the gradient arena after the reduce-scatter only: compared through the parameters)
            for i, (a, b) in enumerate(zip(ga, params_b)):
                assert torch.equal(a, b.grad), (mode, it, "gradient", i, tuple(a.shape))
        for n in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom"):
            assert torch.equal(getattr(den_a, n), getattr(den_b, n)), (mode, it, n)
    if sparse:
        assert arena_a.last_union_fraction is not None and arena_a._order is not None
    need = {"all_reduce", "broadcast"} | ({"all_gather_object"} if cw else set()) | ({"reduce_scatter_tensor", "all_gather_into_tensor"} if mode == "rs_ag" and not sparse or sharded else set())
    assert need <= set(issued), (mode, sparse, sharded, issued)
    print(f"{mode} sparse={sparse} sharded_optimizer={sharded}: 3 steps bit-identical to the path without collectives; issued {issued}", flush=True)
    arena_a.close(); arena_b.close()
torch.cuda.synchronize()
dist.destroy_process_group()
print("rank 0 ok")
'''


def test_rccl_executes_every_collective_shape_of_the_exchange_at_world_size_one(tmp_path):
    script = tmp_path / "rccl_worker.py"
    script.write_text(WORKER)
    ok, msg = run_ranks(script, [ROOT], 1, tmp_path, timeout=900, env_extra={"NCCL_DEBUG": "INFO", "NCCL_DEBUG_FILE": str(tmp_path / "rccl.log")})
    assert ok, msg
    print(msg)
    assert msg.count(" ok") == 1 and "RCCL" in msg
    log = (tmp_path / "rccl.log").read_text() if (tmp_path / "rccl.log").exists() else (tmp_path / "rank0.err").read_text() + (tmp_path / "rank0.out").read_text()
    # the library really initialised a communicator on the device
    assert "NCCL INFO" in log or "RCCL" in log, log[-2000:]
    print("\n".join(l for l in log.splitlines() if "version" in l.lower() or "comm" in l.lower())[:1500])
