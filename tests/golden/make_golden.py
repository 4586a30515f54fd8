"""Generate golden vectors by running the REFERENCE itself (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference (/root/reference) is imported unmodified; weights are our own
seeded synthetic tensors (underwater_image_restoration_amd.synth), loaded via
``load_state_dict(strict=True)`` so the key/shape listing is checked too.
Only inputs' seeds and (sub-sampled) outputs are stored; the reference's
source never enters this repository.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from underwater_image_restoration_amd.synth import (student_param_shapes, synth_state_dict,  # noqa: E402
                                                    synth_student_inputs)
from underwater_image_restoration_amd.synth import daclip_vision_param_shapes, unet_param_shapes  # noqa: E402
from tests.golden.cases import (DACLIP_CASES, STUDENT_BIG_CASES, STUDENT_CASES, STUDENT_ORACLE_CASES, UNET_BIG_CASES,  # noqa: E402
                                UNET_CASES, compress, daclip_image, unet_inputs)


def student_case(name: str, cfg: dict) -> None:
    from underwater_ir.student.naf_unet_wfi import NAFNetWFIGate

    mk = dict(base_channels=cfg["base_channels"], num_levels=3, blocks_per_level=cfg["blocks_per_level"],
              cond_dim=cfg["cond_dim"], num_masks=cfg["num_masks"], num_degradation_types=cfg["num_deg"])
    model = NAFNetWFIGate(**mk).eval()
    shapes = student_param_shapes(**mk)
    ref_shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert set(ref_shapes.keys()) == set(shapes.keys()), "state_dict key-name mismatch"
    assert ref_shapes == dict(shapes), "state_dict shape mismatch"
    sd = synth_state_dict(shapes, seed=cfg["wseed"])
    model.load_state_dict(sd, strict=True)
    x, z, m = synth_student_inputs(cfg["batch"], cfg["H"], cfg["W"], cfg["cond_dim"], cfg["num_masks"],
                                   mask_hw=cfg["mask_hw"], seed=cfg["xseed"])
    taps = {}

    def hook(tag):
        def fn(mod, inp, out):
            if isinstance(out, tuple):
                for i, o in enumerate(out):
                    taps[f"{tag}.{i}"] = o.detach()
            else:
                taps[tag] = out.detach()
        return fn

    for tag in cfg["taps"]:
        model.get_submodule(tag).register_forward_hook(hook(tag))
    with torch.no_grad():
        out, alphas, logits = model(x, z, masks=m)
    store = {"out": out.numpy()}
    store["pre_clamp"] = (taps.pop("head") + x).numpy()
    taps32 = dict(taps)
    taps.clear()
    for i, a in enumerate(alphas):
        store[f"alpha{i}"] = compress(a)
    if logits is not None:
        store["logits"] = logits.numpy()
    for k, v in taps32.items():
        store["tap:" + k] = compress(v)
    # fp64 run of the same reference modules: pins the oracle's MATH to ~1e-12,
    # independent of fp32 summation order.
    model64 = model.double()
    with torch.no_grad():
        o64, a64, l64 = model64(x.double(), z.double(), masks=None if m is None else m.double())
    big = cfg["H"] * cfg["W"] >= 256 * 256   # production size: keep the file small (fp32 roundings of the fp64 run)
    cast = (lambda a: a.astype(np.float32)) if big else (lambda a: a)
    store["out_f64"] = cast(o64.numpy())
    store["pre_clamp_f64"] = cast((taps.pop("head") + x.double()).numpy())
    store["alpha_last_f64"] = compress(a64[-1]).astype(np.float64)
    if l64 is not None:
        store["logits_f64"] = l64.numpy()
    if big:
        del store["pre_clamp"]
    np.savez_compressed(os.path.join(HERE, f"student_{name}.npz"), **store)
    print(name, "out range", float(out.min()), float(out.max()),
          "saturated frac", float(((out == 0) | (out == 1)).float().mean()))


def student_oracle_case(name: str, cfg: dict) -> None:
    """ORACLE-generated vectors (not the reference: see cases.STUDENT_ORACLE_CASES for why): oracle/student_ref.py in fp64.
    Stored: the pre-clamp head output (+ input) rounded to fp32 -- the restored image is its clamp to [0, 1] -- the
    degradation logits and the sub-sampled last alpha map."""
    from oracle import student_ref as R
    mk = dict(base_channels=cfg["base_channels"], num_levels=3, blocks_per_level=cfg["blocks_per_level"],
              cond_dim=cfg["cond_dim"], num_masks=cfg["num_masks"], num_degradation_types=cfg["num_deg"])
    sd = {k: v.double() for k, v in synth_state_dict(student_param_shapes(**mk), seed=cfg["wseed"]).items()}
    x, z, m = synth_student_inputs(cfg["batch"], cfg["H"], cfg["W"], cfg["cond_dim"], cfg["num_masks"],
                                   mask_hw=cfg["mask_hw"], seed=cfg["xseed"])
    taps = {}
    with torch.no_grad():
        out, alphas, logits = R.student_forward(sd, x.double(), z.double(), None if m is None else m.double(),
                                                blocks_per_level=cfg["blocks_per_level"], taps=taps)
    pre = taps["pre_clamp"]
    assert float((pre.clamp(0, 1) - out).abs().max()) == 0.0
    store = {"pre_clamp_f64": pre.numpy().astype(np.float32), "logits_f64": logits.numpy(),
             "alpha_last_f64": compress(alphas[-1]).astype(np.float64), "generated_by": np.array("oracle/student_ref.py fp64")}
    np.savez_compressed(os.path.join(HERE, f"student_{name}.npz"), **store)
    print(name, "out range", float(out.min()), float(out.max()),
          "saturated frac", float(((out == 0) | (out == 1)).double().mean()))


def dwt_case() -> None:
    from underwater_ir.student.naf_unet_wfi.dwt_fft import dwt2, idwt2
    g = torch.Generator().manual_seed(7)
    store = {}
    for tag, shape in (("even", (2, 5, 8, 12)), ("odd", (1, 3, 5, 6))):
        x = torch.randn(shape, generator=g)
        bands = dwt2(x)
        store[f"{tag}_x"] = x.numpy()
        for nm, b in zip(("ll", "lh", "hl", "hh"), bands):
            store[f"{tag}_{nm}"] = b.numpy()
        store[f"{tag}_rec"] = idwt2(*bands).numpy()
    np.savez_compressed(os.path.join(HERE, "dwt.npz"), **store)




def sde_case() -> None:
    """IRSDE tables + short trajectories from the reference's own sde_utils.py (loaded as a single file; its only
    torchvision use is save_image under save_states=True, so an empty stub module satisfies the import)."""
    import importlib.util
    import types
    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tv.utils = tvu
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.utils", tvu)
    spec = importlib.util.spec_from_file_location(
        "ref_sde_utils", "/root/reference/legacy/third_party/universal-image-restoration/utils/sde_utils.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    store = {}
    for sched in ("cosine", "linear", "constant"):
        sde = mod.IRSDE(max_sigma=50, T=100, schedule=sched, eps=0.005)
        for nm in ("thetas", "sigmas", "thetas_cumsum", "sigma_bars"):
            store[f"{sched}_{nm}"] = getattr(sde, nm).numpy()
        store[f"{sched}_dt"] = np.float32(float(sde.dt))
    # trajectories with a closed-form stand-in for the score network and recorded noise
    g = torch.Generator().manual_seed(77)
    mu = torch.rand(2, 3, 8, 8, generator=g)
    sde = mod.IRSDE(max_sigma=50, T=100, schedule="cosine", eps=0.005)
    sde.set_mu(mu)
    sde.set_model(lambda x, mu_, t, **kw: 0.3 * (x - mu_) + 0.01 * torch.sin(x * 7 + float(t)))
    xt = mu + torch.randn(mu.shape, generator=g) * sde.max_sigma
    store["traj_mu"], store["traj_xt"] = mu.numpy(), xt.numpy()
    real_randn_like = torch.randn_like
    for mode in ("sde", "posterior", "ode"):
        tape = []

        def rec(t, *a, **k):
            z = real_randn_like(t, *a, **k)
            tape.append(z.clone())
            return z
        torch.manual_seed(5)
        torch.randn_like = rec
        try:
            if mode == "sde":
                out = sde.reverse_sde(xt, T=6)
            elif mode == "posterior":
                out = sde.reverse_posterior(xt, T=6)
            else:
                out = sde.reverse_ode(xt, T=6)
        finally:
            torch.randn_like = real_randn_like
        store[f"traj_{mode}_out"] = out.numpy()
        if tape:
            store[f"traj_{mode}_tape"] = torch.stack(tape).numpy()
    np.savez_compressed(os.path.join(HERE, "sde.npz"), **store)
    print("sde: dt", float(sde.dt), "sigma_bar[100]", float(sde.sigma_bars[100]))


def _ref_open_clip_transformer():
    """Import the reference's open_clip/transformer.py without the package __init__ (which needs ftfy/torchvision):
    a bare package object with the right __path__, plus a stub for the one torchvision name utils.py imports."""
    import importlib
    import types
    if "open_clip.transformer" in sys.modules:
        return sys.modules["open_clip.transformer"]
    for name in ("torchvision", "torchvision.ops", "torchvision.ops.misc"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision.ops.misc"].FrozenBatchNorm2d = type("FrozenBatchNorm2d", (torch.nn.Module,), {})
    pkg = types.ModuleType("open_clip")
    pkg.__path__ = ["/root/reference/legacy/third_party/universal-image-restoration/open_clip"]
    sys.modules["open_clip"] = pkg
    return importlib.import_module("open_clip.transformer")


def daclip_case(name: str, cfg: dict) -> None:
    """Vision side of DaCLIP assembled from the reference's own classes exactly as daclip_model.py:17-24 does
    (visual, deepcopy -> ControlTransformer), then encode_image(control=True) as daclip_model.py:46-55."""
    import copy
    T = _ref_open_clip_transformer()
    torch.nn.Module.cuda = lambda self, *a, **k: self  # ControlTransformer.__init__ hard-codes .cuda() (transformer.py:297)
    visual = T.VisionTransformer(image_size=cfg["image_size"], patch_size=cfg["patch_size"], width=cfg["width"],
                                 layers=cfg["layers"], heads=cfg["heads"], mlp_ratio=4.0, output_dim=cfg["output_dim"])
    control = copy.deepcopy(visual)
    control.transformer = T.ControlTransformer(control.transformer)
    shapes = daclip_vision_param_shapes(cfg["image_size"], cfg["patch_size"], cfg["width"], cfg["layers"], cfg["output_dim"])
    ref_shapes = {"visual." + k: tuple(v.shape) for k, v in visual.state_dict().items()}
    ref_shapes.update({"visual_control." + k: tuple(v.shape) for k, v in control.state_dict().items()})
    assert ref_shapes == dict(shapes), "DaCLIP vision state_dict mismatch"
    sd = synth_state_dict(shapes, seed=cfg["wseed"])
    visual.load_state_dict({k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")}, strict=True)
    control.load_state_dict({k[len("visual_control."):]: v for k, v in sd.items() if k.startswith("visual_control.")},
                            strict=True)
    visual.eval(), control.eval()
    img = daclip_image(cfg)
    store = {}
    with torch.no_grad():
        store["plain"] = visual(img).numpy()                               # encode_image(control=False)
        degra, hiddens = control(img, output_hiddens=True)
        if cfg["store_hiddens"]:
            for i, h in enumerate(hiddens):                                # LND in the reference -> (B, L, W)
                store[f"hidden{i}"] = h.permute(1, 0, 2).numpy()
        n_h = len(hiddens)
        image = visual(img, control=hiddens)
        assert len(hiddens) == 0 and n_h == cfg["layers"]                  # the list is consumed (popped)
        store["image"], store["degra"] = image.numpy(), degra.numpy()
        v64, c64 = visual.double(), control.double()
        d64, h64 = c64(img.double(), output_hiddens=True)
        store["image_f64"], store["degra_f64"] = v64(img.double(), control=h64).numpy(), d64.numpy()
    np.savez_compressed(os.path.join(HERE, f"daclip_{name}.npz"), **store)
    print("daclip", name, "image |max|", float(np.abs(store["image"]).max()), "degra |max|", float(np.abs(store["degra"]).max()))


def unet_case(name: str, cfg: dict) -> None:
    """ConditionalUNet imported from the reference's daclip-sde app (needs only torch + einops)."""
    sys.path.insert(0, "/root/reference/legacy/third_party/universal-image-restoration/config/daclip-sde")
    from models.modules import ConditionalUNet
    model = ConditionalUNet(3, 3, cfg["nf"], cfg["ch_mult"], 512, cfg["degra"], cfg["image"]).eval()
    shapes = unet_param_shapes(3, 3, cfg["nf"], cfg["ch_mult"], 512, cfg["degra"], cfg["image"])
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == dict(shapes), "UNet state_dict mismatch"
    model.load_state_dict(synth_state_dict(shapes, seed=cfg["wseed"]), strict=True)
    xt, mu, text, image = unet_inputs(cfg)
    taps = {}
    hooks = {"mid": model.mid_block2}
    for i in range(len(cfg["ch_mult"])):
        hooks[f"down{i}"] = model.downs[i][2]
        hooks[f"up{i}"] = model.ups[i][2]
    for tag, mod in hooks.items():
        mod.register_forward_hook(lambda m, inp, out, tag=tag: taps.__setitem__(tag, out.detach()))
    store = {}
    with torch.no_grad():
        for ti, t in enumerate(cfg["times"]):
            out = model(xt, mu, t, text_context=text, image_context=image)
            store[f"out{ti}"] = out.numpy()
            if ti == 0:
                for k, v in taps.items():
                    store["tap:" + k] = compress(v)
        m64 = model.double()
        d = lambda z: None if z is None else z.double()  # noqa: E731
        for ti, t in enumerate(cfg["times"]):
            t64 = torch.tensor([float(t)], dtype=torch.float64)  # an int/float time would give an fp32 embedding
            o64 = m64(xt.double(), mu.double(), t64, text_context=d(text), image_context=d(image)).numpy()
            store[f"out{ti}_f64"] = o64.astype(np.float32) if cfg["H"] * cfg["W"] >= 256 * 256 else o64
    np.savez_compressed(os.path.join(HERE, f"unet_{name}.npz"), **store)
    print("unet", name, "out |max|", float(np.abs(store["out0"]).max()), "fp32-vs-fp64", float(np.abs(store["out0"] - store["out0_f64"]).max()))


def metrics_case() -> None:
    """ssim_loss / psnr_value from the reference's standalone copy (test_ssim.py:10-38; identical to
    train_student.py:70-95, whose package import needs torchvision)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_test_ssim", "/root/reference/test_ssim.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = torch.Generator().manual_seed(41)
    store = {}
    base = torch.rand(2, 3, 40, 56, generator=g)
    cases = {"noisy": (base, (base + 0.05 * torch.randn(base.shape, generator=g)).clamp(0, 1)),
             "random": (base, torch.rand(base.shape, generator=g)),
             "identical": (base, base.clone()),
             "flat": (torch.full((1, 3, 24, 24), 0.5), torch.full((1, 3, 24, 24), 0.5) + 1e-3),
             "small": (torch.rand(1, 1, 7, 5, generator=g), torch.rand(1, 1, 7, 5, generator=g))}
    for name, (a, b) in cases.items():
        store[f"{name}_x"], store[f"{name}_y"] = a.numpy(), b.numpy()
        store[f"{name}_ssim_loss"] = np.float64(float(mod.ssim_loss(a.double(), b.double())))
        store[f"{name}_ssim_loss_f32"] = np.float32(float(mod.ssim_loss(a, b)))
        store[f"{name}_psnr"] = np.float64(mod.psnr_value(a.double(), b.double()))
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **store)
    print("metrics:", {k: float(v) for k, v in store.items() if k.endswith("ssim_loss")})


def imgutil_case() -> None:
    """tensor2img / calculate_psnr of L/utils/img_utils.py.  The module imports cv2 and torchvision.utils.make_grid at
    the top for OTHER functions (save_img, the 4-D grid branch); both are absent here, so empty module objects stand in
    for the import statement only (as SURVEY.md section 8c does for sde_utils.py) -- the two functions used are numpy."""
    import importlib.util
    import types
    for name in ("cv2", "torchvision", "torchvision.utils"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision.utils"].make_grid = None
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    spec = importlib.util.spec_from_file_location(
        "ref_img_utils", "/root/reference/legacy/third_party/universal-image-restoration/utils/img_utils.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = torch.Generator().manual_seed(61)
    a = torch.rand(1, 3, 37, 53, generator=g) * 1.4 - 0.2                 # values outside [0, 1] get clamped
    ties = (torch.arange(0, 256, dtype=torch.float32) + 0.5) / 255.0     # x * 255 lands on .5: round half to even
    a[0, 0, 0, :53] = ties[:53]
    a[0, 1, 1, :53] = ties[100:153]
    b = (a + 0.03 * torch.randn(a.shape, generator=g))
    gray = torch.rand(19, 23, generator=g)
    wide = torch.randn(3, 16, 16, generator=g) * 3
    store = {"a": a.numpy(), "b": b.numpy(), "gray": gray.numpy(), "wide": wide.numpy()}
    ia, ib = mod.tensor2img(a.clone()), mod.tensor2img(b.clone())
    store["a_img"], store["b_img"] = ia, ib
    store["gray_img"] = mod.tensor2img(gray.clone())
    store["wide_img"] = mod.tensor2img(wide.clone(), min_max=(-1, 1))
    store["psnr_ab"] = np.float64(mod.calculate_psnr(ia, ib))
    store["psnr_aa"] = np.float64(mod.calculate_psnr(ia, ia))
    np.savez_compressed(os.path.join(HERE, "imgutil.npz"), **store)
    print("imgutil: psnr", float(store["psnr_ab"]), "shapes", ia.shape, store["gray_img"].shape)


def teacher_inputs(name: str):
    """Seeded inputs of the teacher-head cases (shared with the tests through tests/golden/cases.py would need torch
    only; kept here and stored in the .npz because they are tiny)."""
    g = torch.Generator().manual_seed({"grid7": 51, "crf": 52, "bnc": 53}[name])
    if name == "grid7":      # ViT-B/32 grid, projected 512-d tokens as (B, C, 7, 7); defaults of MaskHead
        b, c, h, w, k = 2, 512, 7, 7, 10
    elif name == "crf":      # non-square grid, CRF on, guidance image at another resolution
        b, c, h, w, k = 2, 64, 6, 9, 5
    else:                    # (B, N, C) tokens with N > C, square grid inferred, CRF with a same-size image
        b, c, h, w, k = 1, 32, 10, 10, 4
    text = torch.randn(k, c, generator=g)
    owner = torch.randint(0, k, (b, h, w), generator=g)
    tok = torch.randn(b, c, h, w, generator=g) + 1.5 * text[owner].permute(0, 3, 1, 2) * torch.rand(b, 1, h, w, generator=g)
    image = None
    if name == "crf":
        image = torch.rand(b, 3, 24, 36, generator=g)
    elif name == "bnc":
        image = torch.rand(b, 3, h, w, generator=g)
        tok = tok.flatten(2).transpose(1, 2).contiguous()
    return tok, text, image


def teacher_case() -> None:
    """MaskHead / DifferentiableGuidedCRF / DegradationCoder (underwater_ir/teacher/*.py), imported unmodified."""
    from underwater_ir.teacher import DegradationCoder, DifferentiableGuidedCRF, MaskHead
    store = {}
    cfgs = {"grid7": dict(ctor={}, call={}),
            "crf": dict(ctor=dict(use_crf=True, temperature=6.0, alpha=3.0, crf_iterations=4, crf_spatial_kernel=3,
                                  crf_spatial_sigma=1.5, crf_edge_weight=4.0, morphology_kernel=3),
                        call=dict(thresholds=[0.2, 0.3, 0.25, 0.5, 0.15], confidence_delta=0.2)),
            "bnc": dict(ctor=dict(use_crf=True, morphology_kernel=5, default_threshold=0.3), call={})}
    for name, cfg in cfgs.items():
        tok, text, image = teacher_inputs(name)
        for dt, sfx in ((torch.float32, ""), (torch.float64, "_f64")):
            head = MaskHead(**cfg["ctor"])
            with torch.no_grad():
                out = head(tok.to(dt), text.to(dt), image=None if image is None else image.to(dt), **cfg["call"])
                code = DegradationCoder()(out, spatial_shape=(3, 4))
            for f in ("scores", "probs", "masks", "intensity", "avg_intensity", "global_prob", "confidence", "raw_probs"):
                store[f"{name}_{f}{sfx}"] = getattr(out, f).numpy()
            for f in ("code", "logits", "confidence_scale"):
                store[f"{name}_{f}{sfx}"] = getattr(code, f).numpy()
            assert code.broadcast.shape == (tok.shape[0], 2 * text.shape[0], 3, 4)
        store[f"{name}_tokens"], store[f"{name}_text"] = tok.numpy(), text.numpy()
        if image is not None:
            store[f"{name}_image"] = image.numpy()
    # the CRF on its own, no confidence weighting
    g = torch.Generator().manual_seed(54)
    probs = torch.softmax(torch.randn(2, 4, 8, 8, generator=g) * 2, dim=1)
    image = torch.rand(2, 3, 8, 8, generator=g)
    with torch.no_grad():
        store["crfonly_out_f64"] = DifferentiableGuidedCRF()(probs.double(), image.double()).numpy()
        store["crfonly_out"] = DifferentiableGuidedCRF()(probs, image).numpy()
    store["crfonly_probs"], store["crfonly_image"] = probs.numpy(), image.numpy()
    np.savez_compressed(os.path.join(HERE, "teacher.npz"), **store)
    print("teacher: mask fill", {n: float(store[f"{n}_masks"].mean()) for n in cfgs},
          "conf", {n: store[f"{n}_confidence"].mean() for n in cfgs})


def enhance_inputs():
    """Seeded inputs of the f3 cases: a smooth colour field + texture + a few saturated / negative pixels."""
    g = torch.Generator().manual_seed(71)
    h, w = 40, 56
    yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
    base = torch.stack([0.25 + 0.5 * xx * yy, 0.55 - 0.3 * xx + 0.2 * torch.sin(6 * yy), 0.6 - 0.4 * yy * yy])
    img = base + 0.08 * torch.randn(3, h, w, generator=g)
    img[:, 3, 5] = 1.3
    img[:, 30, 40] = -0.2
    other = torch.rand(3, h, w, generator=g)
    small = (base[:, ::2, ::2][:, :18, :26] + 0.05 * torch.randn(3, 18, 26, generator=g)).contiguous()
    return img.contiguous(), other, small


def enhance_case() -> None:
    """rl/uiep_cwgr_ops.py, rl/uiqm_uciqe.py, rl/infer_adapt.py imported unmodified; `cv2` is absent here, an empty
    module object satisfies the import statement (none of the functions called below touches it)."""
    import types
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    from underwater_ir.rl import infer_adapt as ia
    from underwater_ir.rl import uiep_cwgr_ops as ops
    from underwater_ir.rl import uiqm_uciqe as qm
    img, other, small = enhance_inputs()
    store = {"img": img.numpy(), "other": other.numpy(), "small": small.numpy()}
    with torch.no_grad():
        store["gamma_0p7"] = ops.gamma_correction(img, 0.7).numpy()
        store["gamma_2p2"] = ops.gamma_correction(img, 2.2).numpy()
        store["gamma_floor"] = ops.gamma_correction(img, -1.0).numpy()
        store["mix_0p3"] = ops.lf_hf_mix(img, other, 0.3).numpy()
        store["mix_1p0"] = ops.lf_hf_mix(img, other, 1.0).numpy()
        store["wd_1p0"] = ops.wavelet_denoise(img).numpy()
        store["wd_2p5"] = ops.wavelet_denoise(img, threshold_scale=2.5).numpy()
        store["wd_f64"] = ops.wavelet_denoise(img.double()).numpy()
        p1, i1 = img[:1].contiguous(), other[:1].contiguous() * 0.5 + img[1:2] * 0.5
        store["gf_p"], store["gf_i"] = p1.numpy(), i1.numpy()
        store["gf_r4"] = ops.guided_filter(p1, i1).numpy()
        store["gf_r2"] = ops.guided_filter(p1, i1, radius=2, eps=1e-2).numpy()
        store["gf_self_f64"] = ops.guided_filter(p1.double(), p1.double()).numpy()
        try:
            ops.guided_filter(img, img)
            store["gf_rgb_raises"] = np.bool_(False)
        except RuntimeError:
            store["gf_rgb_raises"] = np.bool_(True)
        store["msr_small_default"] = ops.multi_scale_retinex(small).numpy()          # 91, 481, 1501-tap kernels
        store["msr_img_s"] = ops.multi_scale_retinex(img, sigmas=(1.5, 4.0)).numpy()
        store["msr_img_f64"] = ops.multi_scale_retinex(img.double(), sigmas=(2.0, 9.0, 15.0)).numpy()
        u8 = qm._image_to_numpy(img)
        store["img_u8"] = u8
        store["uicm"] = np.float64(qm.compute_uicm(u8))
        store["sharpness"] = np.float64(ia.compute_sharpness(img[None]))
        store["sharpness_f64"] = np.float64(ia.compute_sharpness(img[None].double()))
        store["hist_mean"] = np.float64(float(img[None].mean()))
        store["hist_std"] = np.float64(float(img[None].std()))
    np.savez_compressed(os.path.join(HERE, "enhance.npz"), **store)
    print("enhance: uicm", float(store["uicm"]), "sharpness", float(store["sharpness"]), "gf raises on RGB",
          bool(store["gf_rgb_raises"]), "msr mean", float(store["msr_small_default"].mean()))


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = set(sys.argv[1:]) or {"dwt", "student", "sde", "daclip", "unet", "metrics", "teacher", "imgutil", "enhance"}
    if "dwt" in which:
        dwt_case()
    if "student" in which:
        for name, cfg in STUDENT_CASES.items():
            student_case(name, cfg)
    for w in which:                 # one case only: student:NAME
        if w.startswith("student:"):
            student_case(w[8:], STUDENT_CASES[w[8:]])
    if "student_big" in which:      # not in the default set: minutes of CPU time and ~25 GB of RAM (fp64 CFC scores)
        for name, cfg in STUDENT_BIG_CASES.items():
            student_case(name, cfg)
    if "student_oracle" in which:   # 512 x 512 in fp64 through the chunked oracle: ~15 minutes, ~20 GB
        for name, cfg in STUDENT_ORACLE_CASES.items():
            student_oracle_case(name, cfg)
    if "unet_big" in which:
        for name, cfg in UNET_BIG_CASES.items():
            unet_case(name, cfg)
    if "sde" in which:
        sde_case()
    if "daclip" in which:
        for name, cfg in DACLIP_CASES.items():
            daclip_case(name, cfg)
    if "unet" in which:
        for name, cfg in UNET_CASES.items():
            unet_case(name, cfg)
    if "metrics" in which:
        metrics_case()
    if "teacher" in which:
        teacher_case()
    if "imgutil" in which:
        imgutil_case()
    if "enhance" in which:
        enhance_case()
