"""On-disk formats either side of the student path: pseudo-label files and checkpoints.

Mirrors, as host-side logic (no arithmetic):
  * ``load_pseudo_label(root, rel_path)`` -- underwater_ir/student/train_student.py:98-138: ``<stem>.pt`` dict, else the V2
    triple ``<stem>_features.npy / _masks.npy / _probs.npy`` rebuilt into the same dict (confidence 0.8,
    confidence_scale 0.2), else ``FileNotFoundError`` naming every path tried;
  * ``infer_model_dims(pseudo)`` -- train_student.py:368-370 / rl/infer_adapt.py:146-151: cond_dim = z_d.numel(),
    num_masks = masks.shape[0], num_degradation_types = global_prob.numel();
  * ``build_student_from_files`` -- rl/infer_adapt.py:153-160: construct, ``load_state_dict(strict=False)``, eval;
  * ``eval_batch(model, lq, gt, pseudo_list)`` -- the body of evaluate_reference (train_student.py:199-222): stack
    masks / z_d, forward, per-image PSNR / SSIM on clamped outputs, on the device (metrics.batch_metrics).
``.pt`` files are read with ``torch.load(weights_only=True)``: tensors only, nothing from the file is executed.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch


def load_pseudo_label(root, rel_path: str) -> Dict[str, torch.Tensor]:
    base = Path(root) / Path(rel_path)
    pt_path = base.with_suffix(".pt")
    if pt_path.exists():
        return torch.load(pt_path, map_location="cpu", weights_only=True)
    stem, parent = base.stem, base.parent
    f, m, p = (parent / f"{stem}_{k}.npy" for k in ("features", "masks", "probs"))
    if f.exists() and m.exists() and p.exists():
        features = torch.from_numpy(np.load(f, allow_pickle=False)).float()
        masks = torch.from_numpy(np.load(m, allow_pickle=False)).float()
        probs = torch.from_numpy(np.load(p, allow_pickle=False)).float()
        return {"z_d": features, "masks": masks, "global_prob": probs, "confidence": torch.ones_like(probs) * 0.8,
                "confidence_scale": torch.tensor(0.2, dtype=probs.dtype)}
    raise FileNotFoundError(f"Pseudo-label not found for {rel_path}. Tried: {pt_path}, {f}, {m}, {p}")


def infer_model_dims(pseudo: Dict[str, torch.Tensor], num_degradations: Optional[int] = None) -> Tuple[int, int, int]:
    return (int(pseudo["z_d"].numel()), int(pseudo["masks"].shape[0]),
            int(num_degradations or pseudo["global_prob"].numel()))


def build_student_from_files(pseudo: Dict[str, torch.Tensor], checkpoint=None, precision: str = "fp32", device="cuda"):
    """-> (model, z_d (1,cond_dim), masks (1,M,h,w)) ready for ``model(image, z_d, masks=masks)``."""
    from .student import NAFNetWFIGate
    cond_dim, num_masks, num_deg = infer_model_dims(pseudo)
    model = NAFNetWFIGate(cond_dim=cond_dim, num_masks=num_masks, num_degradation_types=num_deg, precision=precision)
    if checkpoint is not None:
        sd = torch.load(checkpoint, map_location="cpu", weights_only=True) if not isinstance(checkpoint, dict) else checkpoint
        model.load_state_dict(sd, strict=False)     # infer_adapt.py:159
    model = model.to(device).eval()
    return model, pseudo["z_d"].reshape(1, -1).to(device), pseudo["masks"].unsqueeze(0).to(device)


@torch.no_grad()
def eval_batch(model, lq: torch.Tensor, gt: torch.Tensor, pseudo: List[Dict[str, torch.Tensor]]):
    """-> (restored, psnr_sum, ssim_sum, count) for one loader batch, as evaluate_reference accumulates them."""
    from . import metrics
    dev = lq.device
    masks = torch.stack([p["masks"] for p in pseudo], dim=0).to(dev)
    z_d = torch.stack([p["z_d"].reshape(-1) for p in pseudo], dim=0).to(dev)
    out, _, _ = model(lq, z_d, masks=masks)
    per = metrics.batch_metrics(out.clamp(0, 1), gt.clamp(0, 1)).double().cpu()
    return out, float(per[:, 2].sum()), float((1.0 - per[:, 0]).sum()), int(per.shape[0])
