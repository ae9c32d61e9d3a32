"""The data-parallel step on the GPU (mdctgan_amd/ddp.py over RCCL; new capability).

* a 1-rank RCCL group drives the real code path on one MI355X: the HIP wgrad kernels fire the reducer's hooks, every
  bucket goes through an RCCL collective, 1/world is folded into Adam -- and the result must equal the plain step bit for
  bit (summing one rank's gradient is the identity);
* the same with reduce-scatter + all-gather buckets (MDCTGAN_DDP_MODE=rs_ag);
* update_fixed_params() (--niter_fix_global) under --fp16 and data parallelism: the new optimiser gets the GradScaler
  slot, a new reducer, grad_scale and pre-step hook;
* 2 ranks on 2 GPUs (skipped on a 1-GPU box): averaged per-rank gradients == single-process gradient on the
  concatenated batch, to float32 rounding.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from oracle import nets as onets

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(*extra, batch=2, netG="global"):
    from mdctgan_amd import options
    from mdctgan_amd.pix2pixHD_model import create_model
    net = ["--netG", netG, "--ngf", "4", "--n_blocks_global", "2", "--n_blocks_attn_g", "0", "--num_D", "2", "--ndf", "8"]
    if netG == "local":
        net += ["--n_downsample_global", "3", "--n_blocks_local", "1"]
    opt = options.make_opt(*options.SPECTRAL_FLAGS, "--lr_sampling_rate", "12000", *net, "--batchSize", str(batch),
                           "--bins", "32", "--segment_length", "7936", "--gpu_ids", "0", *extra)
    m = create_model(opt)
    onets.fill_deterministic(m.netG)
    onets.fill_deterministic(m.netD)
    return m


@pytest.fixture
def one_rank_group():
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        yield
    finally:
        dist.destroy_process_group()


def _batch(golden):
    g = golden("g6_step_global")
    return torch.from_numpy(g["lr"]).to(DEV), torch.from_numpy(g["hr"]).to(DEV)


@pytest.mark.parametrize("mode", ["allreduce", "rs_ag"])
def test_one_rank_rccl_step_is_bit_identical(golden, one_rank_group, mode, monkeypatch):
    from mdctgan_amd import ddp
    monkeypatch.setenv("MDCTGAN_DDP_MODE", mode)
    lr, hr = _batch(golden)
    plain, dp = _model(), _model()
    red = ddp.attach(dp)
    assert red["G"].force and red["G"].mode == mode and len(red["G"].buckets) >= 1
    for _ in range(3):
        lp = plain.optimize_parameters(lr, hr)
        ld = dp.optimize_parameters(lr, hr)
    torch.cuda.synchronize()
    for k in lp:
        assert lp[k].item() == ld[k].item(), k
    for net in ("netG", "netD"):
        for (k, a), (_, b) in zip(getattr(plain, net).state_dict().items(), getattr(dp, net).state_dict().items()):
            assert torch.equal(a, b), (net, k)
    for r in red.values():
        assert r.pending == [b[2] * r.writes_per_step for b in r.buckets]      # every bucket fired and was reset
        r.close()


def test_update_fixed_params_under_fp16_and_ddp(golden, one_rank_group):
    """--niter_fix_global > 0: optimizer_G first holds the local branch only; update_fixed_params() swaps in an
    optimiser over the whole generator.  The --fp16 step after the swap must run (GradScaler slot handed over) and the
    data-parallel wiring must follow the new arena."""
    from mdctgan_amd import ddp
    lr, hr = _batch(golden)
    m = _model("--niter_fix_global", "1", "--fp16", netG="local")
    n_local = sum(p.numel() for k, p in m.netG.named_parameters() if k.startswith("model1"))
    assert sum(p.numel() for p in m.optimizer_G._params) == n_local
    ddp.attach(m)
    m.optimize_parameters(lr, hr)
    before = {k: v.detach().clone() for k, v in m.netG.state_dict().items()}
    old_opt, old_red = m.optimizer_G, m.reducers["G"]
    m.update_fixed_params()
    assert m.optimizer_G is not old_opt and m.reducers["G"] is not old_red
    assert sum(p.numel() for p in m.optimizer_G._params) == sum(p.numel() for p in m.netG.parameters())
    for _ in range(2):
        ld = m.optimize_parameters(lr, hr)
    torch.cuda.synchronize()
    assert m.reducers["G"].flat_g.data_ptr() == m.optimizer_G.flat_g.data_ptr()
    assert m.optimizer_G.pre_step_hook is not None and m.optimizer_G.grad_scale == 1.0
    assert all(np.isfinite(v.item()) for v in ld.values())
    # parameters are unchanged by the swap itself and the GradScaler still holds two slots
    assert len(m.scaler._slots) == 2
    moved = [k for k, v in m.netG.state_dict().items() if not torch.equal(v, before[k])]
    # with these weights the first fp16 iterations may be skipped (scale back-off, fixture G9): only require that nothing
    # blew up and that, if steps were taken, global-branch parameters (not optimised before the swap) moved too
    if moved:
        assert any(k.startswith("model.") for k in moved)
    for r in m.reducers.values():
        r.close()


def test_graphed_step_follows_learning_rate_schedule(golden):
    """update_learning_rate() after capture must reach the replayed Adam kernels: the captured step reads
    lr from the device-resident clock, which the replay wrapper refreshes."""
    lr, hr = _batch(golden)
    eager, graphed = _model(), _model()
    run = graphed.make_graphed_step(lr, hr, warmup=2)
    for _ in range(2):
        eager.optimize_parameters(lr, hr)
    for m in (eager, graphed):
        m.update_learning_rate()
        m.update_learning_rate()
    assert eager.old_lr < 2e-4
    for _ in range(2):
        eager.optimize_parameters(lr, hr)
        run(lr, hr)
    torch.cuda.synchronize()
    for (k, a), (_, b) in zip(eager.netG.state_dict().items(), graphed.netG.state_dict().items()):
        assert torch.equal(a, b), k
    graphed.update_fixed_params()
    with pytest.raises(RuntimeError):
        run(lr, hr)


def _two_rank_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:%d" % rank))
    from mdctgan_amd import ddp
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g6_step_global.npz"))
    lr, hr = torch.from_numpy(g["lr"]).cuda(), torch.from_numpy(g["hr"]).cuda()
    m = _model(batch=1)
    ddp.attach(m)
    m.optimize_parameters(lr[rank:rank + 1], hr[rank:rank + 1])
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({k: v.cpu() for k, v in m.netD.state_dict().items()}, os.path.join(out_dir, "d.pt"))
        torch.save(m.optimizer_G.flat_g.cpu(), os.path.join(out_dir, "gG.pt"))
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_two_rank_rccl_matches_single_process(golden, tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_two_rank_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    lr, hr = _batch(golden)
    single = _model(batch=2)
    single.optimize_parameters(lr, hr)
    g_dp = torch.load(os.path.join(str(tmp_path), "gG.pt")) / 2.0
    g_1 = single.optimizer_G.flat_g.cpu()
    assert (g_dp - g_1).abs().max().item() <= 1e-4 * g_1.abs().max().item()
