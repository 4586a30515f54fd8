"""Headline benchmark: restored megapixels / second of the student restoration forward (BASELINE.json).

    python bench.py [--gpus N --steps K --warmup W] [--precision bf16|fp32] [--batch 16 --size 256]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path over one batch of synthetic images already resident in HBM:
DA-CLIP ViT-B/32 encode_image(control=True) on the (B,3,224,224) CLIP view -> z_d = degra_features (B,512)
-> student NAFNetWFIGate forward on the (B,3,256,256) images conditioned on z_d and (B,10,7,7) masks.
N=1 workload = BASELINE.json configs[1]: batch 16 at 256x256, bf16 MFMA compute.
N>1: images are independent, so every rank runs its own batch (weak scaling, no data-path collective);
RCCL is used only for the barrier and the MAX-over-ranks of the step time.
Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # dense peaks, MI355X_MICROARCH.md


DACLIP_GF_PER_IMAGE = 18.2e9  # both ViT-B/32 towers, SURVEY.md section 6


def student_flops(batch: int, size: int) -> float:
    """462*(S/256)^2 + 1401*(S/256)^4 GF per image (+ 18.2 GF for the DA-CLIP towers)."""
    r = size / 256.0
    return batch * (462e9 * r * r + 1401e9 * r ** 4 + DACLIP_GF_PER_IMAGE)


def cpu_baseline(mk, sd, clip_sd, size: int, threads: int):
    """The oracle (CPU restatement of the reference: oracle/daclip_ref.py + oracle/student_ref.py) timed on the
    host cores on a bounded sample of the same workload (one image)."""
    from oracle import daclip_ref as DR
    from oracle import student_ref as R
    from underwater_image_restoration_amd.synth import synth_student_inputs
    torch.set_num_threads(threads)
    x, _, m = synth_student_inputs(1, size, size, mk["cond_dim"], mk["num_masks"], mask_hw=(7, 7), seed=1234)
    clip_in = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(7))
    t0 = time.time()
    with torch.no_grad():
        _, z = DR.encode_image(clip_sd, clip_in, 12, control=True)
        R.student_forward(sd, x, z, m, blocks_per_level=mk["blocks_per_level"])
    dt = time.time() - t0
    return {"value": round(size * size / 1e6 / dt, 6), "unit": "MP/s", "cores": threads, "kind": "port",
            "sample": f"1 image {size}x{size} fp32: DA-CLIP encode + student forward "
                      f"(oracle/daclip_ref.py + oracle/student_ref.py, torch CPU, {threads} threads), {dt:.1f} s"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--workload", default="student", choices=["student", "tiled1080", "sde", "dataset512"],
                    help="student = BASELINE configs[1] (the headline line); the others are configs[2], [3], [4]")
    ap.add_argument("--sde-steps", type=int, default=100)
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "skip"])
    ap.add_argument("--cpu-size", type=int, default=256)
    ap.add_argument("--no-overlap", action="store_true",
                    help="encode and consumer back to back on one stream (default: encode of step k+1 on a second stream)")
    ap.add_argument("--dual-replica", action="store_true",
                    help="student: two forwards in flight on two buffer sets and streams (pipeline.DualReplicaRestorer; "
                         "+1 %% throughput next to the other two overlaps, -2 %% on the dominant kernel: off by default)")
    ap.add_argument("--serial-branches", action="store_true",
                    help="student: SFFB after WTB on one stream (default: SFFB on a second stream under WTB)")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank rehearsal on ONE GPU: every rank uses cuda:0 and the ranks rendezvous over gloo "
                         "(checks the barrier / max-over-ranks / rank-0 reporting logic; the number is not a result)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from underwater_image_restoration_amd import ops
    from underwater_image_restoration_amd.student import NAFNetWFIGate
    from underwater_image_restoration_amd.daclip import DaCLIPVision
    from underwater_image_restoration_amd.synth import (daclip_vision_param_shapes, student_param_shapes,
                                                        synth_state_dict, synth_student_inputs)

    mk = dict(base_channels=64, num_levels=3, blocks_per_level=2, cond_dim=512, num_masks=10, num_degradation_types=10)
    sd = synth_state_dict(student_param_shapes(**mk), seed=0)
    model = NAFNetWFIGate(**mk, precision=args.precision)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    model.overlap_branches = not args.serial_branches
    clip_sd = synth_state_dict(daclip_vision_param_shapes(), seed=1)
    clip = DaCLIPVision(precision=args.precision)
    clip.load_state_dict(clip_sd)
    clip = clip.cuda().eval()
    x, _, m = synth_student_inputs(args.batch, args.size, args.size, 512, 10, mask_hw=(7, 7), seed=1234 + rank)
    clip_in = torch.randn(args.batch, 3, 224, 224, generator=torch.Generator().manual_seed(7 + rank))
    x, m, clip_in = x.cuda(), m.cuda(), clip_in.cuda()

    # a step = encode(clip view) -> consume(features); U/test.py:114-117: degra_features condition the net
    enc_in = clip_in
    # --dual-replica: two forwards in flight on two activation-buffer sets / streams (pipeline.DualReplicaRestorer)
    from underwater_image_restoration_amd.pipeline import DualReplicaRestorer
    dual = DualReplicaRestorer(model) if args.dual_replica else None

    def consume(img_f, z_d):
        return model(x, z_d, m) if dual is None else dual.submit(x, z_d, m)

    mp_per_step = args.batch * args.size * args.size / 1e6
    wl_name = None
    if args.workload == "tiled1080":     # configs[2]: one 1920x1080 frame through 256-px windows with a 32-px halo
        from underwater_image_restoration_amd.tiling import TiledRestorer
        frame = torch.rand(1, 3, 1080, 1920, generator=torch.Generator().manual_seed(3 + rank)).cuda()
        restorer = TiledRestorer(model, tile=256, halo=32, max_batch=30)
        m1, c1 = m[:1].contiguous(), clip_in[:1].contiguous()

        enc_in = c1

        def consume(img_f, z_d):  # noqa: F811
            return restorer(frame, z_d, m1)
        mp_per_step = 1080 * 1920 / 1e6
        wl_name = (f"BASELINE configs[2]: 1920x1080 frame, 60 halo windows of 256 px (32-px halo), DA-CLIP z_d, "
                   f"{args.precision}")
    elif args.workload == "sde":        # configs[3]: IR-SDE reverse sampling, B=8 at 256x256, T steps
        from underwater_image_restoration_amd.sde import IRSDE
        from underwater_image_restoration_amd.synth import unet_param_shapes
        from underwater_image_restoration_amd.unet import ConditionalUNet
        unet = ConditionalUNet(3, 3, 64, [1, 2, 4, 8], 512, True, True, precision=args.precision)
        unet.load_state_dict(synth_state_dict(unet_param_shapes(), seed=2))
        unet = unet.cuda().eval()
        sde = IRSDE(max_sigma=50, T=100, sample_T=args.sde_steps, schedule="cosine", eps=0.005, device="cuda")
        sde.set_model(unet)
        Bs = 8
        mu = x[:Bs].contiguous()
        sde.set_mu(mu)
        cin = clip_in[:Bs].contiguous()

        enc_in = cin

        def consume(img_f, deg_f):  # noqa: F811
            xt = sde.noise_state(mu)
            return sde.reverse_posterior(xt, text_context=deg_f, image_context=img_f)
        mp_per_step = Bs * args.size * args.size / 1e6
        wl_name = (f"BASELINE configs[3]: IR-SDE reverse_posterior, {args.sde_steps} ConditionalUNet steps, batch {Bs} at "
                   f"{args.size}x{args.size}, DA-CLIP contexts, {args.precision}")
    elif args.workload == "dataset512":  # configs[4]: 512x512 images, sharded over the ranks (weak scaling per step)
        Bd = 4
        xd, _, md = synth_student_inputs(Bd, 512, 512, 512, 10, mask_hw=(7, 7), seed=99 + rank)
        xd, md, cd = xd.cuda(), md.cuda(), clip_in[:Bd].contiguous()

        enc_in = cd

        def consume(img_f, z_d):  # noqa: F811
            return model(xd, z_d, md) if dual is None else dual.submit(xd, z_d, md)
        mp_per_step = Bd * 512 * 512 / 1e6
        wl_name = f"BASELINE configs[4]: 512x512 images, {Bd}/GPU per step, DA-CLIP z_d, {args.precision}"

    # Every step runs BOTH halves inside the timed region.  By default the encode of step k+1 is enqueued on a second
    # HIP stream before the consumer of step k (pipeline.ConditioningPrefetcher): its ~200 small launches fill CUs the
    # consumer's kernels leave idle.  --no-overlap runs the two halves back to back on one stream.
    from underwater_image_restoration_amd.pipeline import ConditioningPrefetcher
    prefetch = None if args.no_overlap else ConditioningPrefetcher(clip)

    def run_steps(k: int) -> None:
        if prefetch is None:
            for _ in range(k):
                consume(*clip.encode_image(enc_in, control=True))
            return
        handle = prefetch.launch(enc_in) if k > 0 else None
        for i in range(k):
            feats = prefetch.take(handle)
            if i + 1 < k:
                handle = prefetch.launch(enc_in)
            consume(*feats)

    run_steps(args.warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ops.event_timing(True)  # HIP events around the main kernel of every flash-attention call (uwir_flash_attn_probe)
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timings = ops.collect_event_timings()
    ops.event_timing(False)
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if args.rehearse else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        mp = world * mp_per_step * args.steps
        # dominant kernel: level-0 WTB flash attention (12 heads x hd 16, n = (S/2)^2): most FLOPs of the path
        n0 = (args.size // 2) ** 2
        dom = [ms for tag, ms in timings if tag[0] == "flash_attn" and tag[2] == n0 and tag[3] == 12]
        roof = None
        traffic = None
        tj = os.path.join(ROOT, "profiles", "r1_v6_pmc_traffic.json")  # rocprofv3 --pmc passes of this same command
        if os.path.exists(tj) and args.batch == 16 and args.size == 256 and args.precision == "bf16":
            with open(tj) as f:
                traffic = json.load(f).get("traffic_bytes_per_launch")
        if dom:
            avg_ms = sum(dom) / len(dom)
            fl = 4.0 * args.batch * 12 * float(n0) ** 2 * (args.size and 16)  # 4*B*heads*n^2*hd
            ach = fl / (avg_ms * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS[args.precision]
            roof = {"bound": "mfma", "kernel": "attn_stream_kernel<hd16,1V,QG4> (WTB level-0 attention, main launch only: HIP events armed "
                              "through uwir_flash_attn_probe on the launch stream)", "achieved": round(ach, 2),
                    "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                    "avg_launch_ms": round(avg_ms, 4), "launches": len(dom)}
        cpu = None
        if args.cpu_baseline != "skip" and world == 1 and args.workload == "student":
            threads = min(16, os.cpu_count() or 1)
            cpu = cpu_baseline(mk, sd, clip_sd, args.cpu_size, threads)
        line = {
            "metric": "restored megapixels/sec (whole node)", "value": round(mp / elapsed, 4),
            "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic" + (" (REHEARSAL: ranks share one GPU)" if args.rehearse else ""),
            "config": {"workload": f"BASELINE configs[1]: DA-CLIP ViT-B/32 encode_image(control=True) -> z_d (512) -> student "
                                   f"NAFNetWFIGate forward, batch {args.batch}/GPU at {args.size}x{args.size}, "
                                   f"{args.precision} MFMA compute, masks (10,7,7)",
                       "batch_per_gpu": args.batch, "size": args.size, "sharding": f"images x{world}",
                       "encode_overlap": "none" if args.no_overlap else "next step's encode on a second HIP stream",
                       "branch_overlap": "none" if args.serial_branches else "SFFB on a second HIP stream under WTB",
                       "forwards_in_flight": 2 if args.dual_replica else 1,
                       "algorithmic_tflop_per_step": round(student_flops(args.batch, args.size) / 1e12, 2),
                       "achieved_tflops_whole_path": round(student_flops(args.batch, args.size) * world * args.steps / elapsed / 1e12, 1)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if wl_name is not None:
            line["config"] = {"workload": wl_name, "sharding": f"independent units x{world}"}
            line["roofline"] = None  # the roofline leg is defined for the headline workload only
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
