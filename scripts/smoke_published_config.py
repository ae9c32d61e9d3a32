"""One optimisation step + one inference with the flags of the reference's train.sh / generate_audio.sh (the published
checkpoints' architecture: netG local, ngf 56, resconv / interpolate sampling, 3 bottleneck-attention blocks of 6 x 128
heads on 8x16 tokens, num_D 3, fit_residual, 16 kHz -> 48 kHz), in float32 and with --fp16."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from mdctgan_amd import options
from mdctgan_amd.pix2pixHD_model import create_model

FLAGS = ["--lr_sampling_rate", "16000", "--sr_sampling_rate", "48000", "--arcsinh_transform", "--abs_spectro",
         "--arcsinh_gain", "1000", "--center", "--norm_range", "-1", "1", "--smooth", "0.0", "--abs_norm", "--src_range", "-5", "5",
         "--netG", "local", "--ngf", "56", "--n_downsample_global", "3", "--n_blocks_global", "4", "--n_blocks_attn_g", "3",
         "--dim_head_g", "128", "--heads_g", "6", "--proj_factor_g", "4", "--n_blocks_attn_l", "0", "--n_blocks_local", "3",
         "--fit_residual", "--upsample_type", "interpolate", "--downsample_type", "resconv", "--num_D", "3", "--lr", "1.5e-4"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for extra in ([], ["--fp16"]):
    torch.manual_seed(0)
    opt = options.make_opt(*FLAGS, *extra, "--batchSize", str(B), "--gpu_ids", "0")
    model = create_model(opt)
    nG = sum(p.numel() for p in model.netG.parameters()); nD = sum(p.numel() for p in model.netD.parameters())
    lr, hr = synth_batch(B, 1, "cuda:0", lr_rate=16000)
    for it in range(3):
        ld = model.optimize_parameters(lr, hr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(5):
        ld = model.optimize_parameters(lr, hr)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    vals = {k: round(float(v), 4) for k, v in ld.items()}
    sr_spectro, sr_audio, *_ = model.inference(lr)
    assert torch.isfinite(sr_audio).all() and all(v == v for v in vals.values())
    print("%s G %.1fM D %.1fM params | %.1f ms/step (eager, batch %d) | losses %s | audio %s" % (
        "fp16" if extra else "fp32", nG / 1e6, nD / 1e6, dt * 1e3, B, vals, tuple(sr_audio.shape)), flush=True)
    del model
    torch.cuda.empty_cache()
