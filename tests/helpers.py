"""Shared helpers for the parity tests (oracle side)."""
from __future__ import annotations

import os

import numpy as np
import torch

from tests.golden.cases import STUDENT_BIG_CASES, STUDENT_CASES, compress
from underwater_image_restoration_amd.synth import student_param_shapes, synth_state_dict, synth_student_inputs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def student_case_setup(name: str):
    cfg = STUDENT_CASES[name] if name in STUDENT_CASES else STUDENT_BIG_CASES[name]
    mk = dict(base_channels=cfg["base_channels"], num_levels=3, blocks_per_level=cfg["blocks_per_level"],
              cond_dim=cfg["cond_dim"], num_masks=cfg["num_masks"], num_degradation_types=cfg["num_deg"])
    sd = synth_state_dict(student_param_shapes(**mk), seed=cfg["wseed"])
    x, z, m = synth_student_inputs(cfg["batch"], cfg["H"], cfg["W"], cfg["cond_dim"], cfg["num_masks"],
                                   mask_hw=cfg["mask_hw"], seed=cfg["xseed"])
    gold = dict(np.load(os.path.join(GOLDEN, f"student_{name}.npz")))
    return cfg, mk, sd, x, z, m, gold


def oracle_tap_lookup(taps: dict, tag: str):
    """Map a reference sub-module path (golden tap name) onto the oracle's taps dict."""
    if tag in ("stem", "final_norm"):
        return taps[tag]
    for suffix, key in ((".hf_block", "hf"), (".lf_block", "lf"), (".cfc.0", "t_out"), (".cfc.1", "f_out"),
                        (".gate.0", "merged"), (".gate.1", "alpha")):
        if tag.endswith(suffix):
            return taps[tag[: -len(suffix)]][key]
    if tag.endswith(".0") and tag[:-2] in taps:
        return taps[tag[:-2]]["out"]
    if tag.endswith(".1") and tag[:-2] in taps:
        return taps[tag[:-2]]["alpha"]
    raise KeyError(tag)


def max_abs(a, b) -> float:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """10*log10(1/mse), capped at 100 dB (train_student.py:91-95 convention)."""
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 100.0 if mse < 1e-10 else 10.0 * float(np.log10(1.0 / mse))


def psnr_delta_at(build: torch.Tensor, ref: torch.Tensor, target_db: float = 30.0, seed: int = 4321) -> float:
    """north_star's bf16 criterion "PSNR delta <= 0.05 dB", made falsifiable: a pseudo ground truth is derived from the
    REFERENCE output by adding Gaussian noise at `target_db` (so the reference scores ~target_db against it, a typical
    restoration quality); returned is |PSNR(ref, gt) - PSNR(build, gt)|.  With an error e independent of that noise
    the delta is 10 log10(1 + mse_e / sigma^2): 0.05 dB at a 30 dB target <=> PSNR(build, ref) >= 49.4 dB, so the
    criterion binds ~20 dB above the old random-target form."""
    g = torch.Generator().manual_seed(seed)
    sigma = 10.0 ** (-target_db / 20.0)
    gt = ref.double().cpu() + sigma * torch.randn(ref.shape, generator=g, dtype=torch.float64)
    return abs(psnr(ref.cpu(), gt) - psnr(build.cpu(), gt))


__all__ = ["psnr_delta_at", "student_case_setup", "oracle_tap_lookup", "max_abs", "psnr", "compress", "STUDENT_CASES"]
