#!/usr/bin/env python
"""bench.py -- the reference's headline metric on MI355X: train steps/s (G+D) on synthetic VCTK-shaped
12 kHz -> 48 kHz segments (BASELINE.json configs[1]: netG=global, ngf 64, 9 ResNet blocks, no attention,
num_D 2, per-GPU batch 8, float32, T = 32512 = 128 frames x 256 bins).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode train|infer]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one train.py:160-202 iteration (forward, 3 discriminator passes, G backward + Adam, D backward +
Adam) through Pix2PixHDModel.optimize_parameters; inputs are resident in HBM before the timed region.
Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- dominant kernel (by event-timed GPU time in the timed region): algorithmic FLOPs per launch /
                  average launch duration (HIP events on the launch stream) vs the dense f32 MFMA peak.
  cpu_baseline -- the CPU oracle's train step (torch CPU, all host cores) on a bounded sample, N == 1 only.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
# multi-process GPU work on this pool: the host driver only supports dmabuf IPC (RCCL / tensor sharing across processes fails
# with hipIpcGetMemHandle otherwise); must be in the environment before the HIP runtime loads
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_F16_MFMA_TFLOPS = 2500.0     # dense f16 MFMA peak (same guide); kernels named "..., 2>" / "..., 3>" compute on it
T_SEG, BATCH = 32512, 8


def synth_batch(batch, seed, device, lr_rate=12000, hr_rate=48000):
    """HR = 0.05 * randn([B, 32512]); LR = HR ideally low-passed to lr_rate/2 (stands in for the resample chain of
    data/audio_dataset.py:66-71).  Built once, outside the timed region."""
    g = torch.Generator().manual_seed(seed)
    hr = 0.05 * torch.randn(batch, T_SEG, generator=g)
    spec = torch.fft.rfft(hr)
    spec[:, int(spec.shape[-1] * lr_rate / hr_rate):] = 0
    lr = torch.fft.irfft(spec, n=T_SEG)
    return lr.to(device), hr.to(device)


class HipEvents:
    """Raw HIP events (ctypes on libamdhip64) handed to the library (mg_probe_arm): its launcher passes them to
    hipExtLaunchKernelGGL for the main GEMM kernel of the conv call, which stamps them with that dispatch's own begin /
    end times on the launch stream -- the duration rocprofv3 reports for the kernel (the older non-DMA launch sites record
    them immediately around their launch instead)."""

    def __init__(self):
        import ctypes
        self.c = ctypes
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]

    def new(self):
        e = self.c.c_void_p()
        assert self.hip.hipEventCreate(self.c.byref(e)) == 0
        return e

    def elapsed_s(self, e0, e1):
        ms = self.c.c_float()
        rc = self.hip.hipEventElapsedTime(self.c.byref(ms), e0, e1)
        return ms.value * 1e-3 if rc == 0 else float("nan")


def is_f16_symbol(name):
    """Which MFMA pipe a GEMM / convolution symbol issues on: hgemm* and the HALF instances of conv_*_dma_kernel<BM, BN, HALF, NBUF>
    (third template argument) the f16 pipe, as do the round-1 conv_*_kernel<..., TAG> instances with TAG bit 1; the rest float32."""
    name = name.rstrip()
    if name.startswith("hgemm"):
        return True
    if "_dma_kernel<" in name:
        return ", true," in name
    return name.startswith("conv_") and name.endswith((", 2>", ", 3>"))


class KernelTimer:
    """Per-kernel GPU time of the conv launches.  mode 'all' (target None): every conv call, used on one warm-up step
    to find the dominant kernel; with a target only that kernel's launches are probed (cheap)."""

    def __init__(self, ops, lib, target=None):
        self.ops, self.lib, self.target, self.records, self.stream_records = ops, lib, target, [], []
        self.ev = HipEvents()
        self.names = {}

    def begin(self, pass_id, g):
        key = (pass_id, g.B, g.H, g.W, g.Ci, g.Co, g.KH, g.stride, g.reflect)
        info = self.names.get(key)
        if info is None:
            info = self.names[key] = (self.ops.plan_name(pass_id, g), self.ops.plan_flops(pass_id, g), self.ops.conv_flops(g))
        if self.target is not None and info[0] != self.target:
            return
        e0, e1 = self.ev.new(), self.ev.new()
        self.lib.mg_probe_arm(e0, e1)
        self.records.append(info + (e0, e1))

    def end(self):
        pass

    # the optimiser's elementwise launches (ops.adam_step_* / scaler_check): events on torch's current stream -- the stream
    # those single-kernel calls are launched on -- right around the launch; the "work" column holds algorithmic HBM bytes
    def stream_begin(self, name, nbytes):
        name = "[hbm] " + name
        if self.target is not None and name != self.target:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        return (name, nbytes, e0, e1)

    def stream_end(self, token):
        token[3].record()
        self.stream_records.append(token)

    def summary(self):
        agg = {}
        for name, kflops, cflops, e0, e1 in self.records:
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += kflops
            a[2] += self.ev.elapsed_s(e0, e1)
            a[3] += cflops
        for name, nbytes, e0, e1 in self.stream_records:
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += nbytes
            a[2] += e0.elapsed_time(e1) * 1e-3
        return agg


CPU_THREADS = 16      # torch CPU threads of the cpu_baseline legs: 16 measured fastest on the 256-core GPU-box host for
                      # these convolutions (8: -10 %, 32: -25 %, 64/128 slower still; scripts/cpu_threads_probe.py)


def _median_time(fn, warmups=3, iters=5):
    """BASELINE.md section 3 protocol: 3 warm-ups, >= 5 timed iterations, median."""
    for _ in range(warmups):
        fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts


def cpu_baseline_train(max_threads):
    """The CPU oracle (oracle/step.py: torch CPU float32 nets + float64 transform == the reference's arithmetic) timed
    on a bounded sample of the same workload: the full configs[1] networks, G+D steps at batch 2 (3 warm-ups + 5 timed,
    median) on a fixed thread count, scaled by 2/8 to the batch-8 step rate (the step is linear in the batch on the
    CPU: convolutions dominate)."""
    from oracle import nets as onets
    from oracle import step as ostep
    gen = torch.Generator().manual_seed(0)
    netG = onets.init_weights(onets.build_generator("global", 2, 1, 64, 4, 9, input_size=(128, 256)), gen)
    netD = onets.init_weights(onets.MultiscaleDRef(3, 64, 3, 2), gen)
    ref = ostep.HotPathRef(netG, netD, ostep.CodecCfg(), num_D=2)
    lr, hr = synth_batch(2, 1, "cpu")
    threads = min(CPU_THREADS, max_threads)
    torch.set_num_threads(threads)
    lrn, hrn = lr.numpy(), hr.numpy()
    med, ts = _median_time(lambda: ref.train_step(lrn, hrn))
    return {"value": round((1.0 / med) * (2.0 / BATCH), 5), "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": "G+D steps of configs[1] at batch 2: 3 warm-ups + 5 timed, median %.2f s (min %.2f, max %.2f), "
                      "scaled x2/8 to batch 8; torch CPU float32 nets + float64 transform on %d threads (fixed; host has "
                      "%d cores)" % (med, ts[0], ts[-1], threads, max_threads)}


def cpu_baseline_train_cfg2(max_threads):
    """configs[2]'s networks (netG=local: 2048-channel trunk + 2 bottleneck-attention blocks, num_D 3) on the CPU oracle:
    G+D steps at batch 1 in float32 (the reference's --fp16 branch is CUDA autocast; on the host the same networks run in
    float32), 1 warm-up + 3 timed, median, scaled x1/8 to the batch-8 step rate."""
    from oracle import nets as onets
    from oracle import step as ostep
    gen = torch.Generator().manual_seed(0)
    netG = onets.init_weights(onets.build_generator("local", 2, 1, 64, 4, 9, 3, input_size=(128, 256), n_attn_g=2, heads_g=8,
                                                    dim_head_g=64), gen)
    netD = onets.init_weights(onets.MultiscaleDRef(3, 64, 3, 3), gen)
    ref = ostep.HotPathRef(netG, netD, ostep.CodecCfg(), num_D=3)
    lr, hr = synth_batch(1, 1, "cpu")
    threads = min(CPU_THREADS, max_threads)
    torch.set_num_threads(threads)
    lrn, hrn = lr.numpy(), hr.numpy()
    med, ts = _median_time(lambda: ref.train_step(lrn, hrn), warmups=1, iters=3)
    return {"value": round((1.0 / med) * (1.0 / BATCH), 5), "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": "G+D steps of configs[2] (float32 on the host) at batch 1: 1 warm-up + 3 timed, median %.2f s (min %.2f, "
                      "max %.2f), scaled x1/8 to batch 8; %d threads (fixed; host has %d cores)" % (med, ts[0], ts[-1], threads, max_threads)}


def _compact(d):
    """A side line as embedded in the headline: without the per-symbol table and the long definition / sample strings (the driver
    keeps the last 8 KB of stdout; round 4's configs[2] --fp16 value was pushed out of it).  The full line is what
    `python bench.py --no-also <that configuration>` prints; the committed copies are under profiles/."""
    d = {k: v for k, v in d.items() if k != "roofline_symbols"}
    for key in ("roofline", "roofline_codec", "cpu_baseline"):
        sub = d.get(key)
        if isinstance(sub, dict):
            d[key] = {k: v for k, v in sub.items() if not k.endswith("_definition") and k not in ("sample", "timed", "note")}
        elif isinstance(sub, list):
            d[key] = [{k: v for k, v in e.items() if not k.endswith("_definition") and k not in ("sample", "timed", "note")}
                      if isinstance(e, dict) else e for e in sub]
    cfg = d.get("config")
    if isinstance(cfg, dict):
        d["config"] = {k: v for k, v in cfg.items() if k != "steps_counted"}
    return d


def also_summary(also):
    """The side lines' values once more, as the LAST key of the headline JSON (inside the final 500 characters of stdout)."""
    out = {}
    for d in also:
        cmd = d.get("cmd", "")
        if "error" in d:
            out["error"] = (out.get("error", "") + " | " + cmd)[:120]
        elif "--config 2" in cmd:
            out["cfg2_fp16_steps_s"], out["cfg2_fp16_ms"] = d.get("value"), d.get("ms_per_step")
        elif "--config 4" in cmd and "MG_F32_SPLIT" in cmd:
            out["cfg4_f32split_audio_s_s"] = d.get("value")
        elif "--config 4" in cmd:
            out["cfg4_audio_s_s"], out["cfg4_ms"] = d.get("value"), d.get("ms_per_step")
            out["cfg4_frac"] = (d.get("roofline") or {}).get("frac")
        elif "--mode codec" in cmd:
            out["codec_clips_s"] = d.get("value")
            rc = d.get("roofline")
            out["codec_frac"] = rc.get("frac") if isinstance(rc, dict) else None
    return out


def also_lines(args):
    """The other BASELINE configurations as their own bench lines, each from a child process of this script (own model, own
    hipGraph, own roofline probe and cpu_baseline), embedded in the headline's JSON so that the driver's one command shows
    them: configs[2] with --fp16 (the reference's configuration for configs[2] / [3]), configs[4] (inference) and the codec pair
    K1 + K2 alone at 4096 clips (north_star's first-named kernels against their HBM roofline)."""
    import subprocess
    out = []
    # (the fourth line is configs[4] once more with MG_F32_SPLIT=1: the Winograd-domain GEMMs as exact three-piece bf16 products --
    # float32-accurate results, tests/test_conv_gpu.py::test_f32_split_gemms_are_float32_accurate -- reported BESIDE the float32-pipe
    # line, never instead of it: DESIGN section 3)
    for extra, env in ((["--config", "2", "--fp16"], None), (["--config", "4"], None), (["--mode", "codec"], None),
                       (["--config", "4", "--no-cpu-baseline"], {"MG_F32_SPLIT": "1"})):
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--no-also"] + extra
        t0 = time.perf_counter()
        label = " ".join("%s=%s" % kv for kv in (env or {}).items())
        label = (label + " " if label else "") + "python bench.py " + " ".join(cmd[2:])
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
            line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1]
            d = _compact(json.loads(line))
            d["wall_s"] = round(time.perf_counter() - t0, 1)
            d["cmd"] = label
        except Exception as e:      # the headline line must survive a failing side line
            d = {"cmd": label, "error": repr(e)[:300]}
        out.append(d)
    return out


def cpu_baseline_infer(max_threads, lr_rate):
    """configs[4]'s path on the CPU oracle: HotPathRef.inference (to_spectro, generator forward, to_audio) on 2 segments
    of 32512 samples, 3 warm-ups + 5 timed, median; audio-s/s = 2 * 32512 / 48000 / median."""
    from oracle import nets as onets
    from oracle import step as ostep
    gen = torch.Generator().manual_seed(0)
    netG = onets.init_weights(onets.build_generator("global", 2, 1, 64, 4, 9, input_size=(128, 256)), gen)
    ref = ostep.HotPathRef(netG, None, ostep.CodecCfg(lr_rate=lr_rate), num_D=2)
    lr, _ = synth_batch(2, 1, "cpu", lr_rate=lr_rate)
    threads = min(CPU_THREADS, max_threads)
    torch.set_num_threads(threads)
    lrn = lr.numpy()
    med, ts = _median_time(lambda: ref.inference(lrn))
    return {"value": round(2 * T_SEG / 48000.0 / med, 4), "unit": "audio-s/s", "cores": threads, "kind": "port",
            "sample": "oracle inference (float64 MDCT + float32 generator + float64 IMDCT) on 2 segments x 32512 samples: "
                      "3 warm-ups + 5 timed, median %.2f s (min %.2f, max %.2f), %d threads (fixed; host has %d cores)"
                      % (med, ts[0], ts[-1], threads, max_threads)}


def live_traffic(child_args, kernels):
    """HBM bytes per launch of `kernels` (short symbol names) MEASURED IN THIS RUN: two rocprofv3 passes -- `--kernel-trace --pmc
    FETCH_SIZE`, then `--pmc WRITE_SIZE`, separate runs as MI355X_MICROARCH.md prescribes -- over a short child process that launches
    those kernels at this line's shapes, reduced like scripts/pmc_traffic.py: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 reports
    half of a wide coalesced read).  Returns {kernel: bytes} for the kernels seen in both passes; {} when rocprofv3 is missing,
    fails or takes longer than 150 s per pass (the caller then keeps the table of profiles/traffic.json and says so)."""
    import csv
    import re
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None or os.environ.get("MG_BENCH_LIVE_PMC", "1") == "0":
        return {}
    # this process is itself running under a profiler (scripts/profile_round.sh): no nested passes
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or "HSA_TOOLS_LIB" in os.environ or any(
            k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return {}

    def short(name):
        return re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").split("(")[0].strip()
    sums = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="mg_pmc_", dir="/tmp")
            cmd = [rp, "--kernel-trace", "--pmc", ctr, "-d", d, "--output-format", "csv", "--", sys.executable] + child_args
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", MG_BENCH_LIVE_PMC="0"), capture_output=True, timeout=150)
            acc, cnt = {}, {}
            for root, _, files in os.walk(d):
                for fn in files:
                    if fn.endswith("counter_collection.csv"):
                        for r in csv.DictReader(open(os.path.join(root, fn))):
                            if r.get("Counter_Name") == ctr:
                                k = short(r["Kernel_Name"])
                                acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
                                cnt[k] = cnt.get(k, 0) + 1
            shutil.rmtree(d, ignore_errors=True)
            sums[ctr] = {k: acc[k] / cnt[k] for k in acc}
    except Exception:       # noqa: BLE001  (a profiler problem must not cost the bench line)
        return {}
    out = {}
    for k in kernels:
        hit = [n for n in sums.get("FETCH_SIZE", {}) if n.startswith(k) and n in sums.get("WRITE_SIZE", {})]
        if hit:
            out[k] = int((2.0 * sums["FETCH_SIZE"][hit[0]] + sums["WRITE_SIZE"][hit[0]]) * 1024)
    return out


LIVE_PMC_NOTE = ("measured in THIS run: separate `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes over `%s` in "
                 "child processes right after the timed region, (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch (gfx950 reports half of "
                 "a wide coalesced read: MI355X_MICROARCH.md, HBM section); `traffic_table` is the value of profiles/traffic.json")


def cpu_baseline_codec(max_threads):
    """The codec pair on the CPU oracle (oracle/transform.py: numpy, float64 transform -- to_spectro + to_audio as K1 + K2 run them)
    on 64 clips of 32512 samples after an 8-clip warm-up: clips/s of the pair."""
    import numpy as np
    from oracle import transform as T
    threads = min(CPU_THREADS, max_threads)
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=threads)
    except Exception:       # noqa: BLE001  (no threadpoolctl: numpy's own default)
        limit, threads = None, max_threads
    w = T.kbd_window(512)
    kw = dict(arcsinh_transform=True, raw_mdct=False, arcsinh_gain=1000.0, abs_norm=True, src_range=(-5.0, 5.0), norm_range=(-1.0, 1.0))
    x = (0.05 * np.random.default_rng(7).standard_normal((64, T_SEG))).astype(np.float32)

    def pair(a):
        s, n = T.to_spectro(a, w, 512, 256, **kw)
        return T.to_audio(s, n, w, 512, 256, **kw)
    pair(x[:8])
    t0 = time.perf_counter()
    pair(x)
    dt = time.perf_counter() - t0
    if limit is not None:
        limit.restore_original_limits()
    return {"value": round(64 / dt, 2), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": "oracle/transform.py to_spectro + to_audio (numpy, float64) on 64 clips x 32512 samples after an 8-clip warm-up: "
                      "%.2f s, BLAS limited to %d threads (host has %d cores)" % (dt, threads, max_threads)}


def bench_codec(args, dev, rank, world):
    """K1 + K2 alone (SURVEY 8d asks for them at a batch where HBM time is measurable): one step = to_spectro
    (pad, frame, window, fold, 256-point DCT-IV on MFMA, arcsinh range-norm) + to_audio (denormalise, inverse DCT,
    window, overlap-add) on --codec-batch clips of 32512 samples.  Algorithmic bytes per clip: K1 reads 130 048 B
    and writes 131 072 B, K2 reads 131 072 B and writes 130 048 B (261 120 B each)."""
    from mdctgan_amd import options
    from mdctgan_amd.pix2pixHD_model import Audio2MDCT
    opt = options.make_opt(*options.SPECTRAL_FLAGS, "--lr_sampling_rate", "12000", "--gpu_ids", str(torch.device(dev).index))
    codec = Audio2MDCT(opt)
    B = args.codec_batch
    g = torch.Generator().manual_seed(7)
    x = (0.05 * torch.randn(B, T_SEG, generator=g)).to(dev)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    def step(e=None):
        with torch.no_grad():
            if e:
                e[0].record()
            spec, pha, norm = codec.to_spectro(x)
            if e:
                e[1].record()
            y = codec.to_audio(spec, norm, pha)
            if e:
                e[2].record()
        return y
    # a step is ~0.6 ms: the W warm-up steps alone end before the clocks have ramped (the first launches of a process read 20-30 %
    # slow, profiles/r04_mdct_ct_ubench.log), so keep warming until 0.25 s of steps have run; the count goes into the line
    n_warm = 0
    t_w = time.perf_counter()
    while n_warm < max(args.warmup, 2) or time.perf_counter() - t_w < 0.25:
        step()
        n_warm += 1
        if n_warm % 16 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(ev[i])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k1 = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps * 1e-3
    k2 = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps * 1e-3
    bytes_k = 261120.0 * B
    flop32 = 2.0 * 128 * 256 * 256 * B
    from mdctgan_amd import _lib
    kn1, kn2 = (_lib.load().mg_mdct_last_kernel(i).decode() for i in (0, 1))      # what the library launched, not a guess
    fam = "ct" if "_ct_" in kn1 else "b3" if "_b3_" in kn1 else "bs"
    # FLOPs the kernels issue per clip: the dense folded 256 x 256 contraction (f32 pipe; x6 as bf16 piece products), or the
    # factored transform's two stages (16 x [16 x 16] + 8 x [32 x 32] matrices per frame = 12 288 multiply-adds)
    flop32 = 2.0 * 128 * 256 * 256 * B
    issued = {"ct": 2.0 * 128 * 12288 * B, "b3": 6.0 * flop32, "bs": flop32}[fam]
    pipe_peak = PEAK_F16_MFMA_TFLOPS if fam == "b3" else PEAK_F32_MFMA_TFLOPS
    out = {"metric": "codec clips/sec (MDCT4+norm, denorm+IMDCT4)", "value": round(world * args.steps * B / dt, 1),
           "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_run": n_warm,
           "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "K1+K2 on %d clips x 32512 samples (128 frames x 256 bins), arcsinh codec" % B,
                      "global_batch": B * world, "segment_length": T_SEG, "parallelism": "dp%d" % world},
           "roofline": {"bound": "hbm", "kernel": "%s / %s" % (kn1, kn2),
                        "achieved": round(bytes_k / k1 / 1e9, 1), "peak": 8000.0,
                        "unit": "GB/s", "frac": round(bytes_k / k1 / 8e12, 4), "traffic": None,
                        "k1_ms": round(k1 * 1e3, 4), "k2_ms": round(k2 * 1e3, 4),
                        "k2_achieved": round(bytes_k / k2 / 1e9, 1), "k2_frac": round(bytes_k / k2 / 8e12, 4),
                        "bytes_per_clip": 261120,
                        "dense_f32_equivalent_tflops_k1": round(flop32 / k1 / 1e12, 2), "dense_f32_equivalent_tflops_k2": round(flop32 / k2 / 1e12, 2),
                        "mfma_pipe": {"ct": "f32 (DCT-IV factored into 8- and 16-point DFT stages: 12 288 multiply-adds per frame instead of 65 536)",
                                      "b3": "bf16 (6 piece products per float32 product)", "bs": "f32"}[fam],
                        "mfma_tflops_k1": round(issued / k1 / 1e12, 2), "mfma_tflops_k2": round(issued / k2 / 1e12, 2),
                        "mfma_frac_k1": round(issued / k1 / 1e12 / pipe_peak, 4),
                        "mfma_frac_k2": round(issued / k2 / 1e12 / pipe_peak, 4),
                        "note": "SURVEY 8d asks for both views.  HBM: 261 120 algorithmic bytes per clip and kernel (frac / k2_frac).  MFMA: "
                                "mfma_* count the FLOPs the launched kernels issue against their pipe's dense peak; dense_f32_equivalent_* "
                                "the folded 256 x 256 contraction (16.8 MFLOP per clip, 64 FLOP per byte) the reference's dense table "
                                "stands for.  With the factored transform (12 FLOP per byte, below the f32 ridge of ~20) the kernels are "
                                "HBM-side; K1 here also returns the mean / std statistics",
                        "timed": "torch events on the launch stream around to_spectro / to_audio (each is one kernel "
                                 "launch plus the output allocation)"}}
    tr = os.path.join(REPO, "profiles", "traffic.json")      # HBM bytes per launch from the rocprofv3 --pmc passes (4096 clips)
    if os.path.exists(tr):
        table = json.load(open(tr)).get("codec", {})
        def first(prefix):
            hits = [v for k, v in table.items() if k.startswith(prefix)]
            return hits[0] if hits else None
        t1, t2 = {"ct": (first("mdct4_ct_kernel<1, true, false, true"), first("imdct4_ct_kernel<1")),
                  "b3": (first("mdct4_b3_kernel<1, true, false, true"), first("imdct4_b3_kernel<1")),
                  "bs": (first("mdct4_bs_kernel<8, 1, false, true"), first("imdct4_bs_kernel<1"))}[fam]
        if t1 and t2:
            out["roofline"]["traffic"] = int(t1 * B / 4096)
            out["roofline"]["k2_traffic"] = int(t2 * B / 4096)
            out["roofline"]["traffic_table"] = [int(t1 * B / 4096), int(t2 * B / 4096)]
            out["roofline"]["traffic_definition"] = (
                "HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 from separate rocprofv3 --pmc passes over this bench "
                "mode at 4096 clips (%s), scaled by clips / 4096" % table.get("_source"))
    if rank == 0 and world == 1 and not args.no_roofline:
        child = [os.path.abspath(__file__), "--mode", "codec", "--codec-batch", str(B), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                 "--no-roofline"]
        live = live_traffic(child, ["mdct4_", "imdct4_"])
        if len(live) == 2:
            out["roofline"]["traffic"], out["roofline"]["k2_traffic"] = live["mdct4_"], live["imdct4_"]
            out["roofline"]["traffic_definition"] = LIVE_PMC_NOTE % "python bench.py --mode codec --steps 3 --warmup 1"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_codec(os.cpu_count() or 1)
    if rank == 0:
        print(json.dumps(out), flush=True)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this script under torch.distributed.run with one rank per GPU
    (the command the driver uses for N > 1), rendezvous on 127.0.0.1 at a free port.  The ranks inherit stdout: rank 0's
    JSON line is still the last line printed.  Returns the launcher's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    rc = subprocess.run(cmd).returncode
    if rc:
        raise SystemExit(rc)


def _wire_bytes(r):
    """Payload one rank hands to the gradient collectives per step (ddp.ArenaReducer): float32 buckets, an explicit 16-bit copy, or
    float32 + float16 pieces where the gradients are stored as float16."""
    if r.wire is not None:
        return int(r.flat_g.numel() * r.wire.element_size())
    total = 0
    for i, (lo, hi, _) in enumerate(r.buckets):
        pieces = r._pieces(i)
        total += sum(t.numel() * t.element_size() for t in pieces) if pieces is not None else (hi - lo) * 4
    return int(total)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="train", choices=["train", "infer", "codec"],
                    help="train: G+D step (the headline).  infer: generate_audio path (inference + segment stitching). "
                         "codec: K1+K2 only (to_spectro + to_audio) on --codec-batch clips, reported against the HBM roofline")
    ap.add_argument("--codec-batch", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="headline line only (no configs[2] --fp16 / configs[4] side lines)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 4],
                    help="BASELINE.json configs index: 1 = netG global (the headline bench line); 2 = netG local + 2 "
                         "bottleneck-attention blocks, num_D 3, run in float32 (the reference config adds --fp16); "
                         "4 = inference, 8k->48k, batch 64 (implies --mode infer)")
    ap.add_argument("--fp16", action="store_true", help="train.py --fp16: autocast convolutions (f16 MFMA) + GradScaler")
    ap.add_argument("--force-ddp", action="store_true", help="run the data-parallel code path even with one rank (testing)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the data-parallel run: nccl = RCCL over xGMI (the product path); gloo "
                         "exists so that the N > 1 launch path can be tested with several ranks sharing one GPU")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)          # plain `python bench.py --gpus N`: start the N ranks ourselves
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started inside a %d-rank launch (WORLD_SIZE): use --gpus %d, or start it "
                         "without a launcher and it spawns its own ranks" % (args.gpus, world, world))
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs an MI355X (torch.cuda.device_count() == 0)")
    if world > n_dev and args.backend == "nccl":
        raise SystemExit("--gpus %d with %d visible GPU(s): RCCL needs one device per rank (--backend gloo shares one GPU "
                         "between ranks for testing the launch path)" % (world, n_dev))
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = "cuda:%d" % dev_index
    use_ddp = world > 1 or args.force_ddp
    if use_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group("gloo")

    from mdctgan_amd import ddp, ops, options
    from mdctgan_amd.pix2pixHD_model import create_model
    torch.manual_seed(42)
    if args.config == 4 and args.mode == "train":
        args.mode = "infer"
    batch = 64 if args.config == 4 else BATCH
    lr_rate = 8000 if args.config == 4 else 12000
    if args.mode == "codec":
        return bench_codec(args, dev, rank, world)
    if args.config in (1, 4):
        net_flags = ["--netG", "global", "--ngf", "64", "--n_downsample_global", "4", "--n_blocks_global", "9",
                     "--n_blocks_attn_g", "0", "--num_D", "2"]
        workload = ("configs[1]: netG=global ngf=64 n_blocks_global=9 n_blocks_attn_g=0 num_D=2, per-GPU batch 8 x 32512 "
                    "samples (128 frames x 256 bins), 12k->48k, fp32")
        if args.config == 4:
            workload = ("configs[4]: generate_audio path, netG=global ngf=64 n_blocks_global=9, 8k->48k, 64 segments x 32512 "
                        "samples per step (model.inference + IMDCT overlap-add + segment stitching), fp32")
    else:
        net_flags = ["--netG", "local", "--ngf", "64", "--n_downsample_global", "4", "--n_blocks_global", "9",
                     "--n_blocks_local", "3", "--n_blocks_attn_g", "2", "--heads_g", "8", "--dim_head_g", "64",
                     "--num_D", "3"]
        workload = ("configs[2] in FLOAT32 (the reference config adds --fp16): netG=local n_blocks_attn_g=2 heads_g=8 "
                    "dim_head_g=64 num_D=3, per-GPU batch 8 x 32512 samples, 12k->48k")
    if args.fp16:
        net_flags = net_flags + ["--fp16"]
        workload = workload.replace("in FLOAT32 (the reference config adds --fp16)", "with --fp16").replace(
            ", fp32", ", --fp16 (autocast convolutions on the f16 MFMA pipe, float32 storage, GradScaler)")
    opt = options.make_opt(*options.SPECTRAL_FLAGS, "--lr_sampling_rate", str(lr_rate), *net_flags,
                           "--batchSize", str(batch), "--gpu_ids", str(dev_index))
    model = create_model(opt)
    if use_ddp:
        ddp.attach(model)
    lr, hr = synth_batch(batch, 42 + rank, dev, lr_rate=lr_rate)      # every rank its own minibatch (weak scaling)
    from mdctgan_amd.generate_audio import generate, make_graphed_generate

    def eager_step():
        if args.mode == "train":
            model.optimize_parameters(lr, hr)
        else:
            generate(model, lr, batch_size=batch, gen_overlap=0)
    step = eager_step
    # data parallel: eager unless MDCTGAN_DDP_GRAPH=1 | auto (auto: every rank tries the capture, all fall back together: Pix2PixHDModel.make_step)
    use_graph = (not args.no_graph) and (not use_ddp or os.environ.get("MDCTGAN_DDP_GRAPH", "0") in ("1", "auto"))

    def fence():
        torch.cuda.synchronize()
        if use_ddp:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up (eager); its last step times every conv launch with HIP events to find the dominant kernel
    from mdctgan_amd import _lib as mglib
    timer_all = KernelTimer(ops, mglib.load())
    # untimed warm-up: at least 6 eager iterations -- the caching allocator and the per-stream workspaces only reach
    # their steady state after a few steps (measured: 27.7 ms/step when timing starts after 2, 19.8 ms after 5)
    nw = max(args.warmup, 6 if not use_graph else 2)
    amp_settle = 0
    if args.mode == "train" and getattr(model, "scaler", None) is not None:
        # --fp16: GradScaler starts at 65536 and halves the scale on every iteration whose float16 gradients overflow -- skipping that
        # iteration's optimiser steps (train.py:183-199).  A skipped step does less work than a real one, so none may sit in the timed
        # region: step (untimed) until the scale has stood still for three iterations, and check again after the timed region.
        still, last = 0, model.scaler.get_scale()
        while still < 3 and amp_settle < 64:
            eager_step()
            amp_settle += 1
            now = model.scaler.get_scale()
            still, last = (still + 1, last) if now == last else (0, now)
        if still < 3:
            raise RuntimeError("the GradScaler's loss scale did not settle in 64 iterations")
    for i in range(nw):
        if i == nw - 1 and not args.no_roofline:
            ops.PROFILER = timer_all
        eager_step()
        ops.PROFILER = None
    torch.cuda.synchronize()
    amp_scale0 = model.scaler.get_scale() if (args.mode == "train" and getattr(model, "scaler", None) is not None) else None
    dominant = None
    symbols = None

    def symbol_table(agg):
        # every GEMM / convolution / optimiser symbol of ONE probed eager step, by time: launches, time per launch, and the rate of the
        # work it was launched with (FLOPs the kernel issues against its MFMA pipe; algorithmic bytes of the "[hbm]" streams)
        rows = []
        for k in sorted(agg, key=lambda k: -agg[k][2])[:12]:
            n, work, secs, _ = agg[k]
            if secs <= 0:
                continue
            if k.startswith("[hbm] "):
                rows.append({"kernel": k[6:], "launches": n, "avg_us": round(secs / n * 1e6, 1), "bound": "hbm",
                             "GBps": round(work / secs / 1e9, 1), "frac": round(work / secs / 8e12, 3)})
            else:
                pk = PEAK_F16_MFMA_TFLOPS if is_f16_symbol(k) else PEAK_F32_MFMA_TFLOPS
                rows.append({"kernel": k, "launches": n, "avg_us": round(secs / n * 1e6, 1), "bound": "mfma",
                             "TFLOPs": round(work / secs / 1e12, 1), "frac": round(work / secs / 1e12 / pk, 3)})
        return rows
    if not args.no_roofline:
        agg = timer_all.summary()
        dominant = max(agg, key=lambda k: agg[k][2]) if agg else None
    timer = KernelTimer(ops, mglib.load(), target=dominant) if dominant else None
    if use_graph and args.mode == "train":
        graphed = model.make_step(lr, hr, warmup=2)           # whole G+D iteration as one hipGraph (data parallel + auto: or eager on every rank)
        step = lambda: graphed()                                # noqa: E731  (inputs already in the captured buffers)
        step()
        if getattr(graphed, "graph", None) is None:
            use_graph = False                                   # the collective fallback took the eager step
    elif use_graph:
        graphed = make_graphed_generate(model, lr, batch_size=batch, gen_overlap=0)   # K1 + generator + K2 + stitch
        step = lambda: graphed()                                # noqa: E731
        step()

    fence()
    if not use_graph:
        ops.PROFILER = timer      # eager: the dominant kernel's launches are event-bracketed inside the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    ops.PROFILER = None
    if amp_scale0 is not None and model.scaler.get_scale() != amp_scale0:
        raise RuntimeError("the loss scale moved inside the timed region (%r -> %r): a skipped optimiser step would have been timed"
                           % (amp_scale0, model.scaler.get_scale()))
    if use_graph and timer is not None:
        # graph replays cannot be bracketed per kernel: time the same launches of the dominant kernel with HIP
        # events over K eager iterations of the same step, immediately after the timed region
        ops.PROFILER = timer
        for _ in range(args.steps):
            eager_step()
        torch.cuda.synchronize()
        ops.PROFILER = None
    if timer is not None and not use_ddp:
        # one more eager step, every launch probed, with the clocks warm: the per-symbol table of the line
        probe_all = KernelTimer(ops, mglib.load())
        ops.PROFILER = probe_all
        eager_step()
        torch.cuda.synchronize()
        ops.PROFILER = None
        symbols = symbol_table(probe_all.summary())
    dist_ranks = None
    if use_ddp:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist_ranks = dist.get_world_size()          # the rank count the process group itself reports

    roofline = None
    if timer is not None and dominant.startswith("[hbm] ") and timer.stream_records:
        # the step's dominant kernel is the optimiser's elementwise stream (configs[2]: 736 M parameters x 30 B): HBM-bound
        torch.cuda.synchronize()
        n, nbytes, secs, _ = timer.summary()[dominant]
        gbps = nbytes / secs / 1e9
        roofline = {"bound": "hbm", "kernel": dominant[6:], "achieved": round(gbps, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(gbps / 8000.0, 4), "traffic": None, "launches": n, "avg_launch_us": round(secs / n * 1e6, 2),
                    "bytes_per_launch": nbytes / n,
                    "bytes_definition": "algorithmic bytes of the Adam update: read p, g, m, v and write p, m, v in float32 (28 B per "
                                        "parameter), + 2 B for the float16 shadow copy under --fp16 (30 B); summed over the launches",
                    "timed": "torch events on the launch stream (torch's current stream, which these single-kernel calls are launched "
                             "on) right around every launch of this kernel in %d eager iterations%s" % (
                                 args.steps, " run right after the graph-replayed timed region" if use_graph else " of the timed region")}
        tr = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tr):
            try:
                section = "configs[%d]%s" % (args.config, " --fp16" if args.fp16 else "")
                doc = json.load(open(tr))
                roofline["traffic"] = (doc.get(section) or {}).get(dominant[6:])
            except Exception:
                pass
    elif timer is not None and timer.records:
        n, flops, secs, conv_flops = timer.summary()[dominant]
        achieved = flops / secs / 1e12
        f16_kernel = is_f16_symbol(dominant)      # float16 GEMMs (conv_h16.h), HALF / TAG bit 1 convolution instances
        peak = PEAK_F16_MFMA_TFLOPS if f16_kernel else PEAK_F32_MFMA_TFLOPS
        split = dominant.startswith("dgemm32g_kernel") and dominant.rstrip().endswith(", 1>")
        if split:      # MG_F32_SPLIT=1: 8 bf16 piece products per float32 product, issued on the bf16 pipe
            flops, achieved, peak = 8.0 * flops, 8.0 * achieved, PEAK_F16_MFMA_TFLOPS
        roofline = {"bound": "mfma", "kernel": dominant, "achieved": round(achieved, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
                    "launches": n, "avg_launch_us": round(secs / n * 1e6, 2),
                    "flops_per_launch": flops / n,
                    **({"f32_split": "MG_F32_SPLIT=1: float32 operands as three exact bf16 pieces each, 8 of the 9 piece products "
                                     "on v_mfma_f32_32x32x16_bf16 with float32 accumulation; achieved / peak count those bf16 "
                                     "MFMA FLOPs (8 x 2*M*N*K) against the dense bf16 peak; in float32-product terms the kernel "
                                     "runs at %.1f TFLOP/s" % (achieved / 8.0)} if split else {}),
                    "flops_definition": "FLOPs the kernel itself issues (2*M*N*K of its GEMM, summed over the launches of this "
                                        "symbol).  dgemm32g_kernel<BM, BN, 2, 2, A, B, 2, P, 0> (csrc/dense_gemm.h) are the batched "
                                        "Winograd-domain GEMMs: layouts (A, B) = (0, 0) forward, (0, 1) data gradient, (1, 1) weight "
                                        "gradient; P = 16 the F(2x2,3x3) layers (the 1024-channel residual trunk), P = 25 the "
                                        "F(2x2,4x4) / F(4x4,2x2) layers; the direct-convolution cost of the same "
                                        "layers (SURVEY 8d, 2*MACs) averages %.3g FLOP per launch" % (conv_flops / n),
                    "timed": ("HIP events stamped with the dispatch's own begin / end times (hipExtLaunchKernelGGL start / stop "
                              "events on the library's launch stream), every launch of this kernel in %d eager iterations run "
                              "right after the graph-replayed timed region" % args.steps) if use_graph else
                             "HIP events stamped with the dispatch's own begin / end times (hipExtLaunchKernelGGL), every launch "
                             "of this kernel inside the timed region"}
        tr = os.path.join(REPO, "profiles", "traffic.json")     # HBM bytes/launch from rocprofv3 --pmc passes
        if os.path.exists(tr):
            try:
                # the same symbol runs on other layer shapes in the other configurations: one section per bench line
                section = "configs[%d]%s" % (args.config, " --fp16" if args.fp16 else "")
                doc = json.load(open(tr))
                table = doc.get(section) or (doc if section == "configs[1]" else {})
                roofline["traffic"] = table.get(dominant)
                roofline["traffic_table"] = table.get(dominant)
                roofline["traffic_definition"] = ("HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 from separate "
                                                  "rocprofv3 --pmc passes on this kernel at this layer shape (gfx950 reports "
                                                  "half of a wide coalesced read: MI355X_MICROARCH.md, HBM section); "
                                                  "profiles/traffic.json section \"%s\", source file in its \"_source\"" % section)
                # a GEMM below the ridge (FLOP per HBM byte < MFMA peak / 8 TB/s) is bounded by its stream, not by the MFMA pipe:
                # say how fast that stream moves (measured traffic of the trunk-layer launch over the launch time of the same symbol)
                tb = roofline["traffic"]
                if tb and roofline.get("flops_per_launch") and roofline.get("avg_launch_us"):
                    ridge = roofline["peak"] * 1e12 / 8e12
                    intensity = roofline["flops_per_launch"] / tb
                    if intensity < ridge:
                        gbps = tb / (roofline["avg_launch_us"] * 1e-6) / 1e9
                        roofline["hbm_view"] = {"bound": "hbm", "flop_per_byte": round(intensity, 1), "ridge": round(ridge, 1),
                                                "traffic_GBps": round(gbps, 1), "peak_GBps": 8000.0, "frac": round(gbps / 8000.0, 4),
                                                "note": "this symbol's intensity is below the ridge: the HBM stream bounds it; bytes = the "
                                                        "measured traffic of its trunk-layer launch (PMC), time = its average launch"}
            except Exception:
                pass
        # the same two PMC passes, live: the dominant kernel's layer of this line through scripts/bench_conv.py in a child process
        layer = {"configs[1]": ["--only", "bottleneck"], "configs[1] --fp16": ["--only", "bottleneck", "--f16"],
                 "configs[2]": ["--only", "trunk2048"], "configs[2] --fp16": ["--only", "trunk2048", "--f16"],
                 "configs[4]": ["--only", "bottleneck_b64"]}.get("configs[%d]%s" % (args.config, " --fp16" if args.fp16 else ""))
        if layer and world == 1 and rank == 0 and dominant.startswith(("dgemm32g", "hgemm", "conv_")):
            torch.cuda.synchronize()
            child = [os.path.join(REPO, "scripts", "bench_conv.py")] + layer + ["--iters", "2"]
            live = live_traffic(child, [dominant])
            if live.get(dominant):
                roofline["traffic"] = live[dominant]
                roofline["traffic_definition"] = LIVE_PMC_NOTE % ("python scripts/bench_conv.py " + " ".join(layer) + " --iters 2")
                if roofline.get("hbm_view") and roofline.get("avg_launch_us"):
                    gbps = live[dominant] / (roofline["avg_launch_us"] * 1e-6) / 1e9
                    roofline["hbm_view"].update({"traffic_GBps": round(gbps, 1), "frac": round(gbps / 8000.0, 4)})

    codec_line = None
    if args.mode == "infer" and not args.no_roofline:
        # the HBM-bound leg of configs[4]: K1 (to_spectro) and K2 (to_audio) on the same 64 segments, torch events on the
        # launch stream around each call, averaged over K eager iterations after the timed region
        pre = model.preprocess
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(max(args.steps, 5))]
        with torch.no_grad():
            for e in evs:
                e[0].record()
                spec, pha, norm = pre.to_spectro(lr)
                e[1].record()
                pre.to_audio(spec, norm, pha)
                e[2].record()
        torch.cuda.synchronize()
        k1 = sum(e[0].elapsed_time(e[1]) for e in evs) / len(evs) * 1e-3
        k2 = sum(e[1].elapsed_time(e[2]) for e in evs) / len(evs) * 1e-3
        bytes_k = 261120.0 * batch
        codec_line = {"bound": "hbm", "k1_us": round(k1 * 1e6, 2), "k2_us": round(k2 * 1e6, 2),
                      "k1_GBps": round(bytes_k / k1 / 1e9, 1), "k2_GBps": round(bytes_k / k2 / 1e9, 1), "peak_GBps": 8000.0,
                      "k2_frac": round(bytes_k / k2 / 8e12, 5), "bytes_per_clip": 261120,
                      "note": "K1 / K2 move 16.7 MB each at batch 64 (2 us at 8 TB/s): launch-latency-dominated at this "
                              "size (SURVEY 8d); bench.py --mode codec reports them at 4096 clips"}

    if rank == 0:
        ms = dt / args.steps * 1e3
        if args.mode == "train":
            metric, value, unit = "train steps/sec (G+D) VCTK 12k->48k", world * args.steps / dt, "steps/s"
        else:
            metric, value, unit = "infer audio-sec/sec", world * args.steps * batch * T_SEG / 48000.0 / dt, "audio-s/s"
        out = {"metric": metric, "value": round(value, 4), "unit": unit, "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f16 products / f32 accumulate (autocast)" if args.fp16 else "f32",
               "data": "synthetic",
               "config": {"workload": workload,
                          "global_batch": batch * world, "segment_length": T_SEG,
                          "parallelism": "dp%d" % world, "launch": "hipGraph replay" if use_graph else "eager",
                          **({"amp": {"loss_scale": amp_scale0, "settle_iterations": amp_settle, "skipped_steps_in_timed_region": 0}}
                             if amp_scale0 is not None else {}),
                          **({"ranks": dist_ranks, "backend": "rccl" if args.backend == "nccl" else args.backend,
                              "ddp_mode": {k: r.mode for k, r in (getattr(model, "reducers", None) or {}).items()},
                              "ddp_wire_bytes_per_step": {k: _wire_bytes(r) for k, r in (getattr(model, "reducers", None) or {}).items()}}
                             if use_ddp else {}),
                          "steps_counted": ("one G+D optimisation step per GPU; value = steps of per-GPU batch 8 completed "
                                            "per second summed over GPUs (weak scaling)") if args.mode == "train" else
                                           "one step = %d segments through inference + stitching per GPU" % batch},
               "roofline": roofline}
        if codec_line is not None:
            out["roofline_codec"] = codec_line
        if symbols:
            out["roofline_symbols"] = {"note": "the 12 heaviest GEMM / convolution / optimiser symbols of ONE eager step probed right after "
                                               "the timed region (the library's own dispatch events; FLOPs = what the kernel issues, "
                                               "2*M*N*K of its GEMM -- the Winograd-domain GEMMs' direct-convolution equivalent is 2.25x)",
                                       "symbols": symbols}
        if world == 1 and not args.no_cpu_baseline and args.mode == "train" and args.config == 1:
            out["cpu_baseline"] = cpu_baseline_train(os.cpu_count() or 1)
        if world == 1 and not args.no_cpu_baseline and args.mode == "train" and args.config == 2:
            out["cpu_baseline"] = cpu_baseline_train_cfg2(os.cpu_count() or 1)
        if world == 1 and not args.no_cpu_baseline and args.mode == "infer":
            out["cpu_baseline"] = cpu_baseline_infer(os.cpu_count() or 1, lr_rate)
    if (rank == 0 and world == 1 and not use_ddp and args.mode == "train" and args.config == 1 and not args.fp16
            and not args.no_also):
        del model
        torch.cuda.empty_cache()
        out["also"] = also_lines(args)
        out["also_summary"] = also_summary(out["also"])
    # the JSON line must be the LAST thing on stdout: RCCL prints a version banner through C stdio, which a pipe only sees
    # when the library's buffer is flushed (normally at exit, i.e. after a line printed here)
    if use_ddp:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if rank == 0:
        if world > 1:
            time.sleep(1.0)          # let the other ranks' processes drain their stdio first
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
