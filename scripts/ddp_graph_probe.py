"""Probe (GPU box, 1 rank): does the data-parallel step capture into a hipGraph with RCCL inside, and does the replay
equal the eager data-parallel step?  Run in its own process: a failed capture can poison the stream state.
    MDCTGAN_DDP_GRAPH=1 python scripts/ddp_graph_probe.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ["MDCTGAN_DDP_GRAPH"] = "1"       # (MDCTGAN_DDP_MODE from the environment: allreduce / rs_ag / sharded)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from mdctgan_amd import ddp, options  # noqa: E402
from mdctgan_amd.pix2pixHD_model import create_model  # noqa: E402
from oracle import nets as onets  # noqa: E402  (deterministic fill only)


def model():
    opt = options.make_opt(*options.SPECTRAL_FLAGS, "--lr_sampling_rate", "12000", "--netG", "global", "--ngf", "4",
                           "--n_blocks_global", "2", "--n_blocks_attn_g", "0", "--num_D", "2", "--ndf", "8",
                           "--batchSize", "2", "--bins", "32", "--segment_length", "7936", "--gpu_ids", "0")
    m = create_model(opt)
    onets.fill_deterministic(m.netG)
    onets.fill_deterministic(m.netD)
    return m


def main():
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g6_step_global.npz"))
    lr, hr = torch.from_numpy(g["lr"]).cuda(), torch.from_numpy(g["hr"]).cuda()
    eager, graphed = model(), model()
    ddp.attach(eager)
    ddp.attach(graphed)
    for _ in range(5):
        eager.optimize_parameters(lr, hr)
    run = graphed.make_graphed_step(lr, hr, warmup=2)
    for _ in range(3):
        run(lr, hr)
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(eager.netG.state_dict().values(), graphed.netG.state_dict().values()))
    print("DDP_GRAPH_PROBE capture=ok replay_equals_eager=%s" % same, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except Exception as e:  # noqa: BLE001
        print("DDP_GRAPH_PROBE capture=failed %s: %s" % (type(e).__name__, str(e)[:300]), flush=True)
        os._exit(3)
