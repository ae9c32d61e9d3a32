"""The reference's command-line contract (options/base_options.py:11-91, options/train_options.py:4-74,
options/audio_config.py:1-13): same flag names, types and defaults, so train.sh / generate_audio.sh flag
sets parse unchanged.  Table-driven; `TrainOptions().parse(args)` returns the same Namespace shape
(`gpu_ids` as a list, `isTrain`), and `make_opt(**overrides)` builds one programmatically.
"""
from __future__ import annotations

import argparse
import os

# options/audio_config.py
N_FFT, HOP_LENGTH, WIN_LENGTH = 512, 256, 512
LR_SAMPLE_RATE, HR_SAMPLE_RATE, SR_SAMPLE_RATE = 8000, 48000, 48000
BINS = 128
assert BINS % 16 == 0
CENTER = True
FRAME_LENGTH = (BINS - 1) * HOP_LENGTH if CENTER else (BINS - 1) * HOP_LENGTH + WIN_LENGTH


def _key_map(x):
    return {str(k): str(v) for k, v in (i.split(":") for i in x.split(","))}


# (flag, kwargs).  `True` as the whole entry means action="store_true".
_FLAGS = [
    # experiment
    ("name", dict(type=str, default="label2city")), ("gpu_ids", dict(type=str, default="0")),
    ("checkpoints_dir", dict(type=str, default="./checkpoints")), ("model", dict(type=str, default="pix2pixHD")),
    ("norm", dict(type=str, default="instance")), ("use_dropout", True),
    ("data_type", dict(type=int, default=32, choices=[8, 16, 32])), ("verbose", True), ("fp16", True),
    ("local_rank", dict(type=int, default=0)), ("seed", dict(type=int, default=42)), ("fit_residual", True),
    # sizes
    ("batchSize", dict(type=int, default=1)), ("loadSize", dict(type=int, default=1024)),
    ("fineSize", dict(type=int, default=512)), ("label_nc", dict(type=int, default=0)),
    ("input_nc", dict(type=int, default=2)), ("output_nc", dict(type=int, default=1)),
    # inputs
    ("dataroot", dict(type=str, default="./datasets/vctk/train.csv")),
    ("evalroot", dict(type=str, default="./datasets/vctk/test.csv")), ("serial_batches", True),
    ("nThreads", dict(type=int, default=2)), ("max_dataset_size", dict(type=int, default=float("inf"))),
    ("explicit_encoding", True), ("alpha", dict(type=float, default=0.6)),
    ("norm_range", dict(type=float, default=(0, 1), nargs=2)), ("abs_norm", True),
    ("src_range", dict(type=float, default=(-5, 5), nargs=2)), ("arcsinh_transform", True), ("raw_mdct", True),
    ("arcsinh_gain", dict(type=float, default=500)), ("add_noise", True), ("snr", dict(type=float, default=55)),
    ("display_winsize", dict(type=int, default=512)), ("tf_log", True),
    # generator
    ("netG", dict(type=str, default="global")), ("ngf", dict(type=int, default=64)),
    ("upsample_type", dict(type=str, default="transconv")), ("downsample_type", dict(type=str, default="conv")),
    ("n_downsample_global", dict(type=int, default=4)), ("n_blocks_global", dict(type=int, default=9)),
    ("n_blocks_attn_g", dict(type=int, default=1)), ("proj_factor_g", dict(type=int, default=4)),
    ("dim_head_g", dict(type=int, default=128)), ("heads_g", dict(type=int, default=4)),
    ("n_blocks_local", dict(type=int, default=3)), ("n_blocks_attn_l", dict(type=int, default=0)),
    ("proj_factor_l", dict(type=int, default=4)), ("dim_head_l", dict(type=int, default=128)),
    ("heads_l", dict(type=int, default=4)), ("n_local_enhancers", dict(type=int, default=1)),
    ("niter_fix_global", dict(type=int, default=0)),
    # masks
    ("mask", True), ("smooth", dict(type=float, default=0.0)), ("mask_hr", True),
    ("mask_mode", dict(type=str, default=None)), ("min_value", dict(type=float, default=1e-7)),
    # train_options: display / bookkeeping
    ("display_freq", dict(type=int, default=200)), ("print_freq", dict(type=int, default=100)),
    ("save_latest_freq", dict(type=int, default=1000)), ("save_epoch_freq", dict(type=int, default=10)),
    ("eval_freq", dict(type=int, default=32000)), ("loss_update_freq", dict(type=int, default=256)),
    ("no_html", True), ("debug", True), ("abs_spectro", True),
    # training
    ("continue_train", True), ("freeze_g_d", True), ("freeze_g_u", True), ("freeze_l_d", True), ("freeze_l_u", True),
    ("load_pretrain", dict(type=str, default="")), ("param_key_map", dict(type=_key_map, default={})),
    ("which_epoch", dict(type=str, default="latest")), ("phase", dict(type=str, default="train")),
    ("niter", dict(type=int, default=100)), ("niter_decay", dict(type=int, default=100)),
    ("niter_limit_aux", dict(type=int, default=20)), ("beta1", dict(type=float, default=0.5)),
    ("lr", dict(type=float, default=0.0002)), ("validation_split", dict(type=float, default=0.05)),
    ("val_indices", dict(type=str)), ("eval_size", dict(type=int, default=100)),
    ("phase_encoding_mode", dict(type=str, default=None)),
    # discriminators
    ("num_D", dict(type=int, default=2)), ("n_layers_D", dict(type=int, default=3)),
    ("ndf", dict(type=int, default=64)), ("no_ganFeat_loss", True), ("lambda_feat", dict(type=float, default=10.0)),
    ("no_lsgan", True), ("pool_size", dict(type=int, default=0)),
    # transform
    ("lr_sampling_rate", dict(type=int, default=LR_SAMPLE_RATE)),
    ("hr_sampling_rate", dict(type=int, default=HR_SAMPLE_RATE)),
    ("sr_sampling_rate", dict(type=int, default=SR_SAMPLE_RATE)),
    ("segment_length", dict(type=int, default=FRAME_LENGTH)), ("gen_overlap", dict(type=int, default=0)),
    ("n_fft", dict(type=int, default=N_FFT)), ("bins", dict(type=int, default=BINS)),
    ("hop_length", dict(type=int, default=HOP_LENGTH)), ("win_length", dict(type=int, default=WIN_LENGTH)),
    ("center", True), ("is_lr_input", True),
]


class TrainOptions:
    def __init__(self):
        self.parser = argparse.ArgumentParser()
        for flag, spec in _FLAGS:
            if spec is True:
                self.parser.add_argument("--" + flag, action="store_true")
            else:
                self.parser.add_argument("--" + flag, **spec)
        self.isTrain = True

    def parse(self, args=None, save=False):
        """args=None parses sys.argv like the reference.  save=True also writes <checkpoints_dir>/<name>/opt.txt
        (base_options.py:117-126); the default here is False so importing code has no filesystem side effect."""
        opt = self.parser.parse_args(args)
        opt.isTrain = self.isTrain
        opt.gpu_ids = [int(s) for s in str(opt.gpu_ids).split(",") if int(s) >= 0]
        if len(opt.gpu_ids) > 0:
            import torch
            if torch.cuda.is_available():
                torch.cuda.set_device(opt.gpu_ids[0])
        if save and not opt.continue_train:
            d = os.path.join(opt.checkpoints_dir, opt.name)
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "opt.txt"), "wt") as f:
                f.write("------------ Options -------------\n")
                for k, v in sorted(vars(opt).items()):
                    f.write("%s: %s\n" % (str(k), str(v)))
                f.write("-------------- End ----------------\n")
        self.opt = opt
        return opt


# the spectral flags every BASELINE config carries (train.sh:9-10)
SPECTRAL_FLAGS = ["--arcsinh_transform", "--abs_spectro", "--arcsinh_gain", "1000", "--norm_range", "-1", "1",
                  "--abs_norm", "--src_range", "-5", "5"]


def make_opt(*flags, **overrides):
    """Namespace from reference-style flags, e.g. make_opt('--netG', 'global', '--batchSize', '8', gpu_ids=[0])."""
    opt = TrainOptions().parse([str(f) for f in flags])
    for k, v in overrides.items():
        setattr(opt, k, v)
    return opt
