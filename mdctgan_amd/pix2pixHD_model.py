"""Model facade with the reference's public surface (models/pix2pixHD_model.py, models/models.py,
models/base_model.py): Audio2MDCT (:14-200), Pix2PixHDModel.{initialize :215-364, forward :394-414,
_forward :416-616, inference :618-638, save, update_learning_rate, update_fixed_params}, InferenceModel,
create_model -- plus optimize_parameters(lr_audio, hr_audio), the G/D step that the reference keeps inline in
train.py:160-202 (BaseModel.optimize_parameters is an empty stub there).

Everything numerical is a HIP launch through the C ABI; torch provides tensors, autograd bookkeeping and the
optimiser / module containers.
"""
from __future__ import annotations

import os
import sys
from typing import Dict

import torch

from . import _lib, networks
from . import amp
from . import functional as Fh
from .mdct import (IMDCT4, MDCT4, _check_geometry, codec_forward, codec_inverse, dct4_table, imdct4_codec, imdct4_generic,
                   kbdwin, mdct4_codec, mdct4_generic)
from .optim import FusedAdam


class Audio2MDCT(torch.nn.Module):
    """models/pix2pixHD_model.py:14-200.  to_spectro == K1 (MDCT + arcsinh + range-norm in one launch),
    to_audio == K2 (denorm + sinh + IMDCT + overlap-add in one launch)."""

    def __init__(self, opt) -> None:
        super().__init__()
        for k, v in vars(opt).items():
            setattr(self, k, v)
        self.device = "cuda" if len(self.gpu_ids) > 0 else "cpu"
        self.up_ratio = self.hr_sampling_rate / self.lr_sampling_rate
        self.window = kbdwin(self.win_length).to(self.device)
        self.min_value = opt.min_value
        self._mdct = MDCT4(n_fft=self.n_fft, hop_length=self.hop_length, win_length=self.win_length,
                           window=self.window, device=self.device)
        self._imdct = IMDCT4(n_fft=self.n_fft, hop_length=self.hop_length, win_length=self.win_length,
                             window=self.window, device=self.device)
        # models/pix2pixHD_model.py:84-106: explicit encoding > arcsinh > raw > dB
        if getattr(self, "explicit_encoding", False):
            self.codec = _lib.MG_CODEC_EXPLICIT
        elif self.arcsinh_transform:
            self.codec = _lib.MG_CODEC_ARCSINH
        elif self.raw_mdct:
            self.codec = _lib.MG_CODEC_RANGE
        else:
            self.codec = _lib.MG_CODEC_DB
        # the hot path (n_fft 512, arcsinh / raw) runs the fused kernels K1 / K2; any other geometry or codec runs the
        # generic kernels of csrc/codec_generic.hip (transform as a dense GEMM + elementwise codec)
        self.geom512 = _check_geometry(self.n_fft, self.hop_length, self.win_length)
        self.fused = self.geom512 and self.codec in (_lib.MG_CODEC_ARCSINH, _lib.MG_CODEC_RANGE)
        self.return_stats = True      # mean / std of norm_param (returned only; costs two atomics per wave)
        self.return_frames = False    # norm_param['frames'] (dead on the hot path; [B, F, 512] of extra traffic)
        self.return_pha = False       # pha = sign(X) * noise is dead on the arcsinh / raw paths

    # -- helpers --------------------------------------------------------------------------------
    def _tables(self, dev):
        if self.window.device != dev:
            self.window = self.window.to(dev)
        return self.window, dct4_table(self.n_fft // 2, dev)

    def _ranges(self):
        return (float(self.norm_range[0]), float(self.norm_range[1])), (float(self.src_range[0]), float(self.src_range[1]))

    def _raw(self, audio, want_frames=False):
        """Raw MDCT coefficients [B, F, M] (+ windowed frames) by the kernel that fits the geometry."""
        if self.geom512:
            window, d4 = self._tables(audio.device)
            r = mdct4_codec(audio, window, d4, self.n_fft, codec=_lib.MG_CODEC_RAW, want_frames=want_frames)
            return r["spec"], r["frames"]
        if self.window.device != audio.device:
            self.window = self.window.to(audio.device)
        return mdct4_generic(audio, self.window, self.n_fft, self.hop_length, True, want_frames)

    def encode(self, audio, want_pair=False, want_stats=None, want_frames=None):
        """K1 (fused) or the generic transform + codec.  Returns the launcher dict (spec [B,F,W] / spec4 [B,C,F,W], pair
        [B,F,W,2] NHWC, min/max, stats)."""
        nr, sr = self._ranges()
        want_stats = self.return_stats if want_stats is None else want_stats
        want_frames = self.return_frames if want_frames is None else want_frames
        if self.fused:
            window, d4 = self._tables(audio.device)
            r = mdct4_codec(audio, window, d4, self.n_fft, codec=self.codec, gain=float(self.arcsinh_gain),
                            norm_range=nr, src_range=sr, per_sample=not self.abs_norm, want_pair=want_pair,
                            want_stats=want_stats, want_frames=want_frames)
            r["spec4"], r["raw"] = r["spec"][:, None], None
            return r
        raw, frames = self._raw(_lib.f32c(audio.reshape(-1, audio.shape[-1])), want_frames)
        r = codec_forward(raw, codec=self.codec, gain=float(self.arcsinh_gain), alpha=float(self.alpha),
                          min_value=float(self.min_value), norm_range=nr, src_range=sr, per_sample=not self.abs_norm,
                          want_pair=want_pair, want_stats=want_stats)
        r["frames"], r["raw"] = frames, raw
        return r

    def _norm_param(self, r, dev):
        if self.abs_norm:
            cache = self.__dict__.setdefault("_range_cache", {})
            if str(dev) not in cache:     # built once: no per-step host->device copy (hipGraph-capturable)
                cache[str(dev)] = (torch.tensor([self.src_range[0]], device=dev)[None, None, None, :],
                                   torch.tensor([self.src_range[1]], device=dev)[None, None, None, :])
            a_min, a_max = cache[str(dev)]
        else:
            a_min, a_max = r["min"].reshape(r["min"].shape[0], -1)[:, :, None, None], r["max"].reshape(r["max"].shape[0], -1)[:, :, None, None]
        mean = std = None
        if r["stats"] is not None:
            n = r["spec4"].numel()
            ms = torch.empty(2, dtype=torch.float32, device=dev)
            _lib.check(_lib.load().mg_stats_finalize(_lib.ptr(r["stats"]), n, _lib.ptr(ms), _lib.stream()), "mg_stats_finalize")
            mean, std = ms[0], ms[1]
        return {"max": a_max, "min": a_min, "mean": mean, "std": std, "frames": r["frames"]}

    # -- reference API --------------------------------------------------------------------------
    def to_spectro(self, audio: torch.Tensor, mask: bool = False, mask_size: int = -1):
        r = self.encode(audio)
        log_spectro = r["spec4"]                              # [B, C, F, W] float32 (C = 2 for --explicit_encoding)
        pha = None
        if self.codec in (_lib.MG_CODEC_DB, _lib.MG_CODEC_EXPLICIT):
            # pix2pixHD_model.py:36, 50-55: the sign of the coefficients, times uniform-rescaled noise unless explicit
            # (to_audio multiplies the dB magnitudes by it).  Random numbers come from torch's generator.
            pha = torch.sign(r["raw"])[:, None]
            if self.codec == _lib.MG_CODEC_DB:
                noise = torch.randn(pha.size(), device=pha.device)
                pha = pha * ((noise - noise.min()) / (noise.max() - noise.min()))
        elif self.return_pha:
            # the normalised value of X = 0: a constant under --abs_norm, per sample from the clip's own (min, max) otherwise
            if self.abs_norm:
                zero = (0.0 - self.src_range[0]) / (self.src_range[1] - self.src_range[0]) * \
                    (self.norm_range[1] - self.norm_range[0]) + self.norm_range[0]
            else:
                mn_b, mx_b = r["min"].reshape(-1, 1, 1, 1), r["max"].reshape(-1, 1, 1, 1)
                zero = (0.0 - mn_b) / (mx_b - mn_b) * (self.norm_range[1] - self.norm_range[0]) + self.norm_range[0]
            noise = torch.randn(log_spectro.size(), device=log_spectro.device)
            noise = (noise - noise.min()) / (noise.max() - noise.min())
            pha = torch.sign(log_spectro - zero) * noise
        if mask:   # pix2pixHD_model.py:57-80 (off in every BASELINE config): overwrite the top bins
            size = log_spectro.size()
            if mask_size == -1:
                mask_size = int(size[3] * (1 - 1 / self.up_ratio))
            if mask_size > 0:
                if self.fit_residual:
                    fill = torch.zeros(size[0], size[1], size[2], mask_size, device=log_spectro.device)
                else:
                    fill = torch.randn(size[0], size[1], size[2], mask_size, device=log_spectro.device)
                    fill = fill / (fill.max() - fill.min())
                log_spectro = torch.cat((log_spectro[:, :, :, :-mask_size], fill), dim=3)
        return log_spectro, pha, self._norm_param(r, audio.device)

    def normalize(self, spectro):
        """API-completeness helper (never on the hot path: to_spectro fuses it into K1).  torch elementwise ops."""
        if self.arcsinh_transform:
            log_spectro = torch.arcsinh(self.arcsinh_gain * spectro) / torch.log(torch.tensor(10.0))
        else:
            log_spectro = spectro
        mean = log_spectro.mean().float()
        std = log_spectro.var().sqrt().float()
        if not self.abs_norm:
            audio_max = log_spectro.flatten(-2).max(dim=-1).values[:, :, None, None].float()
            audio_min = log_spectro.flatten(-2).min(dim=-1).values[:, :, None, None].float()
        else:
            audio_min = torch.tensor([self.src_range[0]], device=spectro.device)[None, None, None, :]
            audio_max = torch.tensor([self.src_range[1]], device=spectro.device)[None, None, None, :]
        log_spectro = (log_spectro - audio_min) / (audio_max - audio_min)
        log_spectro = log_spectro * (self.norm_range[1] - self.norm_range[0]) + self.norm_range[0]
        return log_spectro, audio_max, audio_min, mean, std

    def denormalize(self, log_spectro: torch.Tensor, min: torch.Tensor, max: torch.Tensor):
        """API-completeness helper (to_audio fuses it into K2)."""
        x = (log_spectro - self.norm_range[0]) / (self.norm_range[1] - self.norm_range[0])
        x = x * (max - min) + min
        if self.arcsinh_transform:
            return torch.sinh(x * torch.log(torch.tensor(10.0))) / self.arcsinh_gain
        return x

    def to_audio(self, log_spectro: torch.Tensor, norm_param: Dict[str, torch.Tensor], pha: torch.Tensor = None, stitch=None):
        """pix2pixHD_model.py:139-165.  stitch = (out, gen_overlap, first_seg) (fused geometry only): K2 writes the segments
        straight into the stitched waveform `out` (generate_audio.py:40-53 inside the kernel) and `out` is returned."""
        if stitch is not None and not self.fused:
            raise NotImplementedError("stitched decode needs the fused 512 / 256 geometry")
        nr, sr = self._ranges()
        mn, mx = norm_param["min"], norm_param["max"]
        per_sample = mn.numel() > 1
        if not per_sample:
            sr = (float(mn.reshape(-1)[0]), float(mx.reshape(-1)[0])) if not self.abs_norm else sr
        if self.fused:
            window, d4 = self._tables(log_spectro.device)
            spec = log_spectro.squeeze(1) if log_spectro.dim() == 4 else log_spectro
            audio, _ = imdct4_codec(spec, window, d4, self.n_fft, codec=self.codec, gain=float(self.arcsinh_gain),
                                    norm_range=nr, src_range=sr, min_b=mn if per_sample else None,
                                    max_b=mx if per_sample else None, stitch=stitch)
            return audio if stitch is not None else audio[:, None, None, :]
        spec4 = log_spectro if log_spectro.dim() == 4 else log_spectro[:, None]
        raw = codec_inverse(spec4, codec=self.codec, gain=float(self.arcsinh_gain), alpha=float(self.alpha),
                            min_value=float(self.min_value), norm_range=nr, src_range=sr,
                            min_b=mn if per_sample else None, max_b=mx if per_sample else None)
        if self.codec == _lib.MG_CODEC_DB and pha is not None and self.up_ratio > 1:
            # pix2pixHD_model.py:147-157: the sign restore sits INSIDE `if self.up_ratio > 1` in the reference (at
            # up_ratio == 1 its dB decode stays unsigned) -- mirrored as written
            ph = pha.reshape(raw.shape).to(raw.device)
            size = ph.size(-2)
            keep = int(size * (1 / self.up_ratio))
            pseudo = (2 * torch.randint(low=0, high=2, size=ph.size(), device=raw.device) - 1).to(ph.dtype)
            ph = torch.cat((ph[..., :keep, :], pseudo[..., keep:, :]), dim=-2)
            raw = raw * ph
        if self.geom512:
            window, d4 = self._tables(raw.device)
            audio, _ = imdct4_codec(raw, window, d4, self.n_fft, codec=_lib.MG_CODEC_RAW)
        else:
            if self.window.device != raw.device:
                self.window = self.window.to(raw.device)
            audio, _ = imdct4_generic(raw, self.window, self.n_fft, self.hop_length, True)
        return audio[:, None, None, :]

    def forward(self, lr_audio: torch.Tensor):
        with torch.no_grad():
            return self.to_spectro(lr_audio, mask=self.mask)

    def hr_forward(self, hr_audio: torch.Tensor):
        with torch.no_grad():
            return self.to_spectro(hr_audio, mask=self.mask_hr,
                                   mask_size=int(self.n_fft * (1 - self.sr_sampling_rate / self.hr_sampling_rate) // 2))


class BaseModel(torch.nn.Module):
    """models/base_model.py (save / load helpers keep the reference's file naming and fall-backs)."""

    def name(self):
        return "BaseModel"

    def initialize(self, opt):
        self.opt = opt
        self.gpu_ids = opt.gpu_ids
        self.isTrain = opt.isTrain
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self.device = "cuda" if len(self.gpu_ids) > 0 else "cpu"

    def save_network(self, network, network_label, epoch_label, gpu_ids=None):
        save_path = os.path.join(self.save_dir, "%s_net_%s.pth" % (epoch_label, network_label))
        os.makedirs(self.save_dir, exist_ok=True)
        # plain contiguous NCHW tensors on the host: loads in the reference unchanged
        sd = {k: v.detach().to("cpu").contiguous() for k, v in network.state_dict().items()}
        torch.save(sd, save_path)

    def load_network(self, network, network_label, epoch_label, save_dir=""):
        save_path = os.path.join(save_dir or self.save_dir, "%s_net_%s.pth" % (epoch_label, network_label))
        if not os.path.isfile(save_path):
            print("%s not exists yet!" % save_path)
            if network_label == "G":
                raise FileNotFoundError("Generator must exist!")
            return
        pretrained = torch.load(save_path, map_location="cpu")
        try:
            network.load_state_dict(pretrained)
        except RuntimeError:
            model_dict = network.state_dict()
            usable = {k: v for k, v in pretrained.items() if k in model_dict and v.size() == model_dict[k].size()}
            module_map = getattr(self.opt, "param_key_map", {}) or {}
            for name, param in pretrained.items():      # base_model.py:72-89 key remapping
                parts = name.split(".")
                key = ".".join(parts[:2])
                if name not in usable and key in module_map:
                    parts[1] = module_map[key]
                    mapped = ".".join(parts)
                    if mapped in model_dict and param.size() == model_dict[mapped].size():
                        usable[mapped] = param
            missing = [k for k in model_dict if k not in usable]
            if missing and getattr(self.opt, "verbose", False):
                print("Pretrained network %s: %d tensors not initialised" % (network_label, len(missing)))
            network.load_state_dict(usable, strict=False)


class Pix2PixHDModel(BaseModel):
    def name(self):
        return "Pix2PixHDModel"

    # -- construction ---------------------------------------------------------------------------
    def initialize(self, opt):
        BaseModel.initialize(self, opt)
        for k, v in vars(opt).items():
            setattr(self, k, v)
        self.isTrain = opt.isTrain
        # --fp16 (train.py:65-70): autocast arithmetic for the convolutions + one device-side GradScaler
        self.fp16 = bool(getattr(opt, "fp16", False))
        self.scaler = amp.GradScaler(device=self.device if hasattr(self, "device") else "cuda") if (
            self.fp16 and self.isTrain) else None
        input_nc = opt.label_nc if opt.label_nc != 0 else opt.input_nc
        self.preprocess = Audio2MDCT(opt)
        self.preprocess.return_stats = False
        self.freeze = opt.freeze_g_d or opt.freeze_g_u or opt.freeze_l_d or opt.freeze_l_u
        self.netG = networks.define_G(
            input_nc, opt.output_nc, opt.ngf, opt.netG, opt.n_downsample_global, opt.n_blocks_global,
            opt.n_local_enhancers, opt.n_blocks_local, opt.norm, gpu_ids=self.gpu_ids,
            upsample_type=opt.upsample_type, downsample_type=opt.downsample_type,
            input_size=(opt.bins, opt.n_fft // 2), n_attn_g=opt.n_blocks_attn_g, n_attn_l=opt.n_blocks_attn_l,
            proj_factor_g=opt.proj_factor_g, heads_g=opt.heads_g, dim_head_g=opt.dim_head_g,
            proj_factor_l=opt.proj_factor_l, heads_l=opt.heads_l, dim_head_l=opt.dim_head_l)
        self.netG.set_freeze(opt.freeze_g_d, opt.freeze_g_u, opt.freeze_l_d, opt.freeze_l_u)
        if self.isTrain:
            if opt.no_lsgan and not opt.no_ganFeat_loss:
                # with feature matching on, NLayerDiscriminator.forward never applies its Sigmoid (networks.py:684-689) and
                # nn.BCELoss rejects the logits ("all elements of input should be between 0 and 1"): the reference stops here too
                raise NotImplementedError("--no_lsgan needs --no_ganFeat_loss (the reference's BCELoss would be fed logits)")
            self.netD = networks.define_D(input_nc + opt.output_nc, opt.ndf, opt.n_layers_D, opt.norm, bool(opt.no_lsgan),
                                          opt.num_D, not opt.no_ganFeat_loss, gpu_ids=self.gpu_ids)
        if not self.isTrain or opt.continue_train or opt.load_pretrain:
            self.load_network(self.netG, "G", opt.which_epoch, opt.load_pretrain)
            if self.isTrain:
                self.load_network(self.netD, "D", opt.which_epoch, opt.load_pretrain)
        if self.isTrain:
            from .image_pool import ImagePool
            self.fake_pool = ImagePool(opt.pool_size)          # pix2pixHD_model.py:294-298 (default 0: no history)
            self.old_lr = opt.lr
            self.limit_aux_loss = False
            self.criterionGAN = networks.GANLoss(use_lsgan=not opt.no_lsgan, device=self.device)
            self.loss_names = ["G_GAN"] + ([] if opt.no_ganFeat_loss else ["G_GAN_Feat"]) + ["D_real", "D_fake"]
            if opt.niter_fix_global > 0:
                params = [v for k, v in self.netG.named_parameters()
                          if k.startswith("model" + str(opt.n_local_enhancers))]
            else:
                params = list(self.netG.parameters())
            shadow = bool(getattr(opt, "fp16", False))      # autocast: Adam keeps the float16 weight operands current
            self.optimizer_G = FusedAdam([p for p in params if p.requires_grad], lr=opt.lr, betas=(opt.beta1, 0.999),
                                         half_shadow=shadow)
            self.optimizer_D = FusedAdam(list(self.netD.parameters()), lr=opt.lr, betas=(opt.beta1, 0.999),
                                         half_shadow=shadow)
        # A D pass whose weight gradients train.py discards (the G-loss pass) skips its wgrad launches; set False
        # to reproduce the reference's wasted work bit for bit.
        self.skip_discarded_d_grads = True
        self.stack_d_loss_passes = True      # D(fake.detach()) and D(real) run as one pass over 2B samples
        # ... and D(fake) of the G loss reuses the fake half of that pass (same weights, same input: identical
        # activations) -- one discriminator forward per iteration, two backward passes through it (Fh.backward_pass)
        self.share_d_fake_pass = os.environ.get("MDCTGAN_SHARE_D_PASS", "1") != "0"
        self._shared_rows = 0
        self.current_lable = self.current_generated = self.current_real = None

    def loss_filter(self, g_gan, g_gan_feat, d_real, d_fake):
        out = [g_gan]
        if not self.no_ganFeat_loss:
            out.append(g_gan_feat)
        return out + [d_real, d_fake]

    # -- forward --------------------------------------------------------------------------------
    def _two_channel(self, spectro):
        if self.abs_spectro and self.arcsinh_transform:
            return Fh.g_input(spectro, float(self.norm_range[0]))
        return spectro

    def forward(self, lr_audio, hr_audio):
        """pix2pixHD_model.py:394-414."""
        self._finish_pending()
        pre = self.preprocess
        lr_spectro, lr_pha, lr_norm_param = pre.forward(lr_audio)
        hr_spectro, hr_pha, hr_norm_param = pre.hr_forward(hr_audio)
        lr_input = self._two_channel(lr_spectro)
        sr_spectro = self.netG.forward(lr_input)
        if self.fit_residual:
            sr_spectro = Fh.add(sr_spectro, lr_spectro)
        return sr_spectro, None, hr_spectro, hr_pha, hr_norm_param, lr_spectro, lr_pha, lr_norm_param

    def _d_in(self, lr_spectro, x_spectro):
        if self.abs_spectro and self.arcsinh_transform:
            return Fh.d_input(lr_spectro, x_spectro, float(self.norm_range[0]))
        return Fh.cat_channels(lr_spectro, x_spectro)        # pix2pixHD_model.py:425-427, 440: sr_input = sr_spectro

    def _forward(self, lr_audio, hr_audio, infer=False, share_d_pass=False):
        """pix2pixHD_model.py:416-616: the four live losses [G_GAN, G_GAN_Feat, D_real, D_fake].  share_d_pass: the
        caller backpropagates the two losses under Fh.backward_pass (optimize_parameters does); otherwise every loss
        keeps its own discriminator pass and plain loss.backward() calls work as in train.py."""
        sr_spectro, _, hr_spectro, _, hr_norm_param, lr_spectro, _, lr_norm_param = self.forward(lr_audio, hr_audio)
        pooled = self.isTrain and getattr(self.opt, "pool_size", 0) > 0     # the fake pass of the D loss sees the history
        stacked = (self.stack_d_loss_passes and self.abs_spectro and self.arcsinh_transform and not pooled
                   and not self.no_lsgan and not self.no_ganFeat_loss)      # the stacked-pass loss kernels are the LSGAN + feature ones
        shared = (share_d_pass and stacked and self.share_d_fake_pass and self.skip_discarded_d_grads
                  and torch.is_grad_enabled() and sr_spectro.requires_grad)
        self._shared_rows = 0
        B = lr_spectro.shape[0]
        if shared:
            # ONE discriminator forward over [fake, real]: D loss from both halves, G loss from the fake half
            self._shared_rows = B
            pred_both = self.netD.forward(Fh.d_input_shared(lr_spectro, sr_spectro, hr_spectro,
                                                            float(self.norm_range[0])), weight_grad="D0")
        elif stacked:
            # D(fake.detach()) and D(real) as ONE pass over a batch of 2B: every layer of the discriminator is
            # per-sample (InstanceNorm), so stacking is exact; half the launches, twice the rows per GEMM
            pred_both = self.netD.forward(Fh.d_input_pair(lr_spectro, sr_spectro, hr_spectro,
                                                          float(self.norm_range[0])))
        if stacked:       # the sums over the scales are one node each: the kernels accumulate, no elementwise adds
            loss_D_fake, loss_D_real = Fh.mse_const_pair_loss_sum([scale_out[-1] for scale_out in pred_both],
                                                                  self.criterionGAN.fake_label, self.criterionGAN.real_label)
        else:
            fake_in = self._d_in(lr_spectro, sr_spectro.detach())
            if pooled:                                # discriminate_F(..., use_pool=True), pix2pixHD_model.py:366-374
                fake_in = self.fake_pool.query(fake_in).contiguous(memory_format=torch.channels_last)
            pred_fake_pool = self.netD.forward(fake_in)
            loss_D_fake = self.criterionGAN(pred_fake_pool, False)
            pred_real = self.netD.forward(self._d_in(lr_spectro, hr_spectro))
            loss_D_real = self.criterionGAN(pred_real, True)
        feat_weights = 4.0 / (self.n_layers_D + 1)
        D_weights = 1.0 / self.num_D
        loss_G_GAN_Feat = 0
        if shared:
            loss_G_GAN = Fh.mse_const_first_half_loss_sum([scale_out[-1] for scale_out in pred_both],
                                                          self.criterionGAN.real_label)
            if not self.no_ganFeat_loss:
                loss_G_GAN_Feat = Fh.l1_halves_loss_sum([t for scale_out in pred_both for t in scale_out[:-1]],
                                                        D_weights * feat_weights * self.lambda_feat)
        else:
            if stacked:
                pred_real = [[t.detach()[B:] for t in scale_out] for scale_out in pred_both]
            pred_fake = self.netD.forward(self._d_in(lr_spectro, sr_spectro),
                                          weight_grad=not self.skip_discarded_d_grads)
            loss_G_GAN = self.criterionGAN(pred_fake, True)
            if not self.no_ganFeat_loss:
                for i in range(self.num_D):
                    for j in range(len(pred_fake[i]) - 1):
                        loss_G_GAN_Feat = loss_G_GAN_Feat + Fh.l1_loss(
                            pred_fake[i][j], pred_real[i][j].detach(), D_weights * feat_weights * self.lambda_feat)
        # visuals are materialised lazily (the reference copies three tensors to the host every step)
        self._visual_src = (lr_spectro, sr_spectro.detach(), hr_spectro, lr_norm_param, hr_norm_param)
        return [self.loss_filter(loss_G_GAN, loss_G_GAN_Feat, loss_D_real, loss_D_fake),
                torch.empty if not infer else sr_spectro]

    def optimize_parameters(self, lr_audio, hr_audio):
        """One train.py:160-202 iteration (float32 branch): forward, G step, D step.  Returns the loss dict
        (device scalars; call .item() only when you need to print)."""
        self._finish_pending()
        with amp.autocast(self.fp16):
            losses, _ = self._forward(lr_audio, hr_audio, infer=False, share_d_pass=True)
        loss_dict = dict(zip(self.loss_names, losses))
        # train.py:170-171, 183, 194: loss_D = (D_fake + D_real) * 0.5, loss_G = G_GAN + G_GAN_Feat, scaler.scale(loss).  The constant
        # factors (0.5, the loss scale S) ride in the gradient each backward pass starts from -- d loss_D / d(D_fake + D_real) =
        # 0.5 S either way, bit for bit -- instead of in elementwise launches on one-element tensors (3 multiplies forward, 2 backward,
        # 2 ones_like fills per iteration)
        loss_D = loss_dict["D_fake"] + loss_dict["D_real"]
        loss_G = loss_dict["G_GAN"] + loss_dict["G_GAN_Feat"] if "G_GAN_Feat" in loss_dict else loss_dict["G_GAN"]
        red = getattr(self, "reducers", None)      # data-parallel gradient reducers (mdctgan_amd.ddp.attach)
        sc = self.scaler                           # train.py:183-199: one GradScaler, updated once per iteration
        seed_G, seed_D = self._backward_seeds(sc, loss_D.device)
        if red:
            red["G"].active, red["D"].active = True, False
        rows = self._shared_rows      # > 0: both losses hang off one discriminator forward (see _forward)
        g_kw = dict(retain_graph=True) if rows else {}
        d_kw = dict(inputs=[p for p in self.netD.parameters() if p.requires_grad]) if rows else {}
        self.optimizer_G.zero_grad()
        with Fh.backward_pass("G" if rows else None, rows), Fh.fused_adam_scope(self.optimizer_G if sc is None and not red else None):
            loss_G.backward(gradient=seed_G, **g_kw)
            if sc is not None:
                sc.step(self.optimizer_G)
            else:
                self.optimizer_G.step()
        if red:
            red["G"].active, red["D"].active = False, True
        self.optimizer_D.zero_grad()
        with Fh.backward_pass("D" if rows else None, rows):
            loss_D.backward(gradient=seed_D, **d_kw)
            if sc is not None:
                sc.step(self.optimizer_D)
                sc.update()
            else:
                self.optimizer_D.step()
        if red and getattr(self, "ddp_check_steps", 0) > 0 and not torch.cuda.is_current_stream_capturing():
            # the first data-parallel steps: every rank's parameter arenas must equal rank 0's bit for bit (ddp.check_replicas)
            from . import ddp
            self.ddp_check_steps -= 1
            self._finish_pending()
            ddp.check_replicas(self.optimizer_G.flat_p, red["G"].group, "generator")
            ddp.check_replicas(self.optimizer_D.flat_p, red["D"].group, "discriminator")
        # detached: the caller only prints / logs these; handing out the graph would keep one generator's worth of
        # saved activations alive until the next iteration overwrites the dict
        return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in loss_dict.items()}

    def _backward_seeds(self, sc, dev):
        """(d loss_G, d loss_D) the two backward passes start from: (1, 0.5), times the GradScaler's loss scale under --fp16 (read from
        its device state when the loss-gradient kernels run, so a captured step follows the scale)."""
        cache = self.__dict__.setdefault("_seed_cache", {})
        if dev not in cache:
            cache[dev] = (torch.ones((), device=dev), torch.full((), 0.5, device=dev))
        one, half = cache[dev]
        if sc is None or not sc.enabled:
            return one, half
        s = sc.state[0]
        return s, s * 0.5

    def _finish_pending(self):
        """Sharded data parallelism (ddp "sharded" mode): the all-gather of the last update may still be in flight."""
        for name in ("optimizer_G", "optimizer_D"):
            opt_ = getattr(self, name, None)
            if opt_ is not None:
                opt_.finish_pending()

    def make_graphed_step(self, lr_audio, hr_audio, warmup=3, _ddp_ok=False, _fail_in_capture=False):
        """Capture one full optimize_parameters() iteration (~470 launches: forward, both backward passes, both Adam
        steps) into a hipGraph and return run(lr, hr) -> loss dict, which copies the batch into the captured input
        buffers and replays.  The optimiser clock and learning rate live in HBM, so replays advance Adam exactly
        like eager steps.  The warm-up iterations are real training steps."""
        if getattr(self, "reducers", None) and not _ddp_ok and os.environ.get("MDCTGAN_DDP_GRAPH", "0") != "1":
            # RCCL collectives are capturable (ProcessGroupNCCL records them into the graph from its own stream), but
            # that path has only been exercised with a 1-rank group on this hardware pool: opt-in
            raise NotImplementedError("graph capture of the data-parallel step is opt-in: MDCTGAN_DDP_GRAPH=1")
        if getattr(self.opt, "pool_size", 0) > 0:
            # ImagePool.query draws random.uniform / randint on the host: a replay would repeat the captured choices, and the
            # stored history would live in the graph's private pool where replays overwrite it
            raise NotImplementedError("--pool_size > 0 takes host-side random decisions every step: not capturable")
        static_lr, static_hr = lr_audio.clone(), hr_audio.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 2)):
                self.optimize_parameters(static_lr, static_hr)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # Under data parallelism ProcessGroupNCCL's watchdog thread polls events while this thread captures: in the default
        # "global" capture mode any such call from another thread invalidates the capture ("operation failed due to a previous
        # error during capture": seen on 2 of 5 fresh boxes in the first RCCL process, where start-up is slowest).  Only this
        # thread's calls belong to the capture.
        mode = "thread_local" if getattr(self, "reducers", None) else "global"
        with torch.cuda.graph(graph, capture_error_mode=mode):
            if _fail_in_capture:       # test hook (make_step, MDCTGAN_DDP_GRAPH_FAIL_RANK): after the warm-up steps every rank took part in
                raise RuntimeError("simulated capture failure (MDCTGAN_DDP_GRAPH_FAIL_RANK)")
            losses = self.optimize_parameters(static_lr, static_hr)

        captured = (self.optimizer_G, self.optimizer_D)

        def run(lr=None, hr=None):
            if (self.optimizer_G, self.optimizer_D) != captured:
                raise RuntimeError("an optimiser was replaced after capture (update_fixed_params): "
                                   "call make_graphed_step() again")
            # the captured launches read lr from the device-resident Adam clock; FusedAdam.step() -- which refreshes
            # it -- is not called again after capture, so pick up update_learning_rate() here
            self.optimizer_G.sync_lr()
            self.optimizer_D.sync_lr()
            if lr is not None:
                static_lr.copy_(lr, non_blocking=True)
            if hr is not None:
                static_hr.copy_(hr, non_blocking=True)
            graph.replay()
            Fh.bump_weight_epoch()            # the replay ran both Adam steps
            return losses
        run.graph = graph
        return run

    def make_step(self, lr_audio, hr_audio, warmup=3):
        """run(lr, hr) -> loss dict for a training loop: the captured step (make_graphed_step) where that is safe, eager
        optimize_parameters() otherwise.  run.graph is the hipGraph or None.
          * single process: captured;
          * data parallel, MDCTGAN_DDP_GRAPH unset / "0": eager (the default: a capture with RCCL collectives inside has only ever
            run with a 1-rank group on this hardware pool -- DESIGN section 5);
          * MDCTGAN_DDP_GRAPH=1: captured, a capture error propagates;
          * MDCTGAN_DDP_GRAPH=auto: every rank tries the capture; one MAX all-reduce of "it failed here" before the first replay
            makes ALL ranks fall back to eager steps when ANY rank's capture failed -- eight ranks cannot strand each other with
            seven replaying a graph (whose collectives wait for the eighth) and one stepping eagerly."""
        import torch.distributed as dist
        red = getattr(self, "reducers", None)
        how = os.environ.get("MDCTGAN_DDP_GRAPH", "0") if red else "1"

        def eager(lr=None, hr=None):
            return self.optimize_parameters(lr_audio if lr is None else lr, hr_audio if hr is None else hr)
        eager.graph = None
        if how == "0":
            return eager
        if how == "1":
            return self.make_graphed_step(lr_audio, hr_audio, warmup=warmup, _ddp_ok=True)
        if how != "auto":
            raise ValueError("MDCTGAN_DDP_GRAPH must be 0, 1 or auto")
        run, failed = None, 0
        try:
            # test hook: this rank's capture "fails" -- AFTER the warm-up iterations (real training steps whose bucket all-reduces
            # every rank must take part in), where a real capture error would strike
            fail_rank = os.environ.get("MDCTGAN_DDP_GRAPH_FAIL_RANK")
            fail_here = fail_rank is not None and dist.is_initialized() and (fail_rank == "all" or dist.get_rank() == int(fail_rank))
            run = self.make_graphed_step(lr_audio, hr_audio, warmup=warmup, _ddp_ok=True, _fail_in_capture=fail_here)
        except Exception as e:      # noqa: BLE001 -- whatever the capture raised, the decision is collective
            failed = 1
            print("[mdctgan_amd] hipGraph capture of the data-parallel step failed on this rank (%s): asking every rank to step "
                  "eagerly" % (repr(e)[:200],), file=sys.stderr, flush=True)
        if dist.is_initialized():
            flag = torch.tensor([float(failed)], device=lr_audio.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=red["G"].group)
            failed = int(flag.item())
        if failed:
            # EVERY rank drops what the attempt left behind before its first eager step -- the rank whose capture broke holds
            # collective handles of the aborted capture and half-counted buckets, the others a graph nobody will replay
            run = None
            torch.cuda.synchronize()
            for r in red.values():
                r.abort()
            for opt_ in (self.optimizer_G, self.optimizer_D):
                opt_._pending = None
                opt_.resync_shadow()
            return eager
        return run

    def inference(self, lr_audio, stitch=None):
        """pix2pixHD_model.py:618-638.  stitch: see Audio2MDCT.to_audio (sr_audio is then the stitched waveform)."""
        self._finish_pending()
        with torch.no_grad():
            lr_spectro, lr_pha, lr_norm_param = self.preprocess.forward(lr_audio)
            sr_spectro = self.netG.forward(self._two_channel(lr_spectro))
            if self.fit_residual:
                lr_part = int(sr_spectro.size(-1) / self.preprocess.up_ratio)
                sr_spectro[..., :lr_part] *= 1e-3
                sr_spectro = Fh.add(sr_spectro, lr_spectro)
            sr_audio = self.preprocess.to_audio(sr_spectro, lr_norm_param, lr_pha, stitch=stitch)
        return sr_spectro, sr_audio, lr_pha, lr_norm_param, lr_spectro

    # -- bookkeeping ----------------------------------------------------------------------------
    def save(self, which_epoch):
        self._finish_pending()
        self.save_network(self.netG, "G", which_epoch, self.gpu_ids)
        self.save_network(self.netD, "D", which_epoch, self.gpu_ids)

    def update_fixed_params(self):
        """pix2pixHD_model.py:640-646 (after --niter_fix_global epochs: optimise the whole generator).  The new optimiser
        owns a new arena, so everything that was wired to the old one follows: the GradScaler's found_inf slot, and under
        data parallelism the gradient reducer, the 1/world gradient scale and the pre-step hook."""
        old = self.optimizer_G
        # sharded data parallelism: the previous step's all-gather into the OLD parameter arena may still be in flight on
        # RCCL's stream; the new arena copies from views of the old one, so wait (and re-cast the old float16 shadow) first
        old.finish_pending()
        params = [p for p in self.netG.parameters() if p.requires_grad]
        for p in params:                       # the new arena copies the current values; gradients start fresh
            p.grad = None
        self.optimizer_G = FusedAdam(params, lr=self.lr, betas=(self.beta1, 0.999), half_shadow=old.half_shadow)
        if self.scaler is not None:
            self.scaler.release(old)
        red = getattr(self, "reducers", None)
        if red:
            from . import ddp
            red["G"].close()
            red["G"] = ddp.attach_optimizer(self.optimizer_G, 1, red["G"].bucket_bytes, red["G"].group)
            red["G"].active = True

    def update_learning_rate(self):
        lrd = self.lr / self.niter_decay
        lr = self.old_lr - lrd
        for opt_ in (self.optimizer_D, self.optimizer_G):
            for param_group in opt_.param_groups:
                param_group["lr"] = lr
        self.old_lr = lr

    def get_current_visuals(self):
        """Raw denormalised spectrogram arrays of sample 0 (pix2pixHD_model.py:569-590).  The matplotlib rendering
        of util/spectro_img.py is observability, not part of the hot path."""
        lr_s, sr_s, hr_s, lr_n, hr_n = self._visual_src
        nr0, nr1 = self.norm_range

        def den(s, n):
            return ((s[0, 0].float().cpu() - nr0) / (nr1 - nr0) * (n["max"].reshape(-1)[0].cpu() - n["min"].reshape(-1)[0].cpu())
                    + n["min"].reshape(-1)[0].cpu()).numpy()
        return {"lable_spectro": den(lr_s, lr_n), "generated_spectro": den(sr_s, lr_n), "real_spectro": den(hr_s, hr_n)}


class InferenceModel(Pix2PixHDModel):
    def forward(self, lr_audio):
        return self.inference(lr_audio)


def create_model(opt):
    """models/models.py:3-20."""
    if opt.model != "pix2pixHD":
        raise NotImplementedError("only --model pix2pixHD exists on the hot path")
    model = Pix2PixHDModel() if opt.isTrain else InferenceModel()
    model.initialize(opt)
    return model
