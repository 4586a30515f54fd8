"""Golden-vector case table shared by make_golden.py (reference side) and the tests (oracle / HIP side)."""
from __future__ import annotations

import numpy as np

_BLOCK0 = "encoder_stages.0.blocks.0"
_TAPS = ["head", "stem", _BLOCK0 + ".hf_block", _BLOCK0 + ".lf_block", _BLOCK0 + ".cfc", _BLOCK0 + ".gate",
         _BLOCK0, "encoder_stages.2.blocks.0", "decoder_stages.1.blocks.0", "final_norm"]

STUDENT_CASES = {
    # full-width model (head_dim 16/32/64 like production), DA-CLIP style cond (512) + (K,7,7) masks
    "full32": dict(base_channels=64, blocks_per_level=2, cond_dim=512, num_masks=10, num_deg=10, batch=1, H=32, W=32,
                   mask_hw=(7, 7), wseed=0, xseed=1234, taps=_TAPS),
    # non-square (power-of-two sides: see oracle note on self-conjugate FFT bins), batch 2, SigLIP2-style cond (1024) + (10,224,224) masks, one block per level
    "rect": dict(base_channels=64, blocks_per_level=1, cond_dim=1024, num_masks=10, num_deg=None, batch=2, H=64, W=32,
                 mask_hw=(224, 224), wseed=1, xseed=99, taps=_TAPS),
    # V1 teacher style cond (20), no masks, narrow net (head_dim 4/8/16)
    "narrow": dict(base_channels=16, blocks_per_level=1, cond_dim=20, num_masks=0, num_deg=5, batch=2, H=64, W=64,
                   mask_hw=(7, 7), wseed=2, xseed=5, taps=_TAPS),
    # even sides whose deeper levels are odd: 38 -> 19 -> (dwt2 drops a row: 18) -> 9 -> (8), 44 -> 22 -> 11 -> (10).  The
    # first block of a stage shrinks its odd input to the even size below it (dwt_fft.py:38 is a stride-2 valid conv) and the
    # decoder resizes the up-sampled map to the skip's size (model.py:150-151).  FFT lines of 19 / 11 / 9 / 5 / 4 bins.
    "evenodd": dict(base_channels=16, blocks_per_level=2, cond_dim=20, num_masks=3, num_deg=5, batch=2, H=38, W=44,
                    mask_hw=(7, 7), wseed=6, xseed=17, taps=_TAPS),
}

DACLIP_CASES = {
    # reduced width: every stage exercised, outputs + all hiddens stored in full
    "small": dict(image_size=64, patch_size=32, width=128, layers=3, heads=2, output_dim=64, batch=3, wseed=3, xseed=21,
                  store_hiddens=True),
    # 7x7 grid (50 tokens like ViT-B/32) at reduced width, more layers than control pops in a symmetric position
    "grid7": dict(image_size=224, patch_size=32, width=192, layers=4, heads=3, output_dim=512, batch=2, wseed=4, xseed=22,
                  store_hiddens=True),
    # the production tower: daclip_ViT-B-32.json (width 768, 12 layers, 12 heads), outputs only
    "vitb32": dict(image_size=224, patch_size=32, width=768, layers=12, heads=12, output_dim=512, batch=1, wseed=5,
                   xseed=23, store_hiddens=False),
}


def daclip_image(cfg: dict):
    import torch
    g = torch.Generator().manual_seed(cfg["xseed"])
    return torch.randn(cfg["batch"], 3, cfg["image_size"], cfg["image_size"], generator=g)

UNET_CASES = {
    # LinearAttention only (depth 2), odd sizes -> reflect pad + crop, integer time
    "lin": dict(nf=32, ch_mult=[1, 2], degra=True, image=False, batch=2, H=30, W=36, times=[50], wseed=6, xseed=31),
    # the test.yml topology at half width: SpatialTransformer at level 3 + mid, both contexts, two times
    "full4": dict(nf=32, ch_mult=[1, 2, 4, 8], degra=True, image=True, batch=2, H=64, W=48, times=[100, 1], wseed=7,
                  xseed=32),
    # production width (U/options/test.yml:30-39), float time like t * sample_scale
    "prod": dict(nf=64, ch_mult=[1, 2, 4, 8], degra=True, image=True, batch=1, H=32, W=32, times=[37.0], wseed=8,
                 xseed=33),
    # no conditioning branches at all (context unused), ch_mult with a repeated last width
    "noctx": dict(nf=32, ch_mult=[1, 2, 4, 4], degra=False, image=False, batch=1, H=32, W=32, times=[7], wseed=9,
                  xseed=34),
}


def unet_inputs(cfg: dict):
    """mu = x ~ U[0,1), xt = mu + randn * 50/255, contexts ~ N(0,1)."""
    import torch
    g = torch.Generator().manual_seed(cfg["xseed"])
    mu = torch.rand(cfg["batch"], 3, cfg["H"], cfg["W"], generator=g)
    xt = mu + torch.randn(mu.shape, generator=g) * (50 / 255)
    text = torch.randn(cfg["batch"], 512, generator=g)
    image = torch.randn(cfg["batch"], 512, generator=g)
    return xt, mu, (text if cfg["degra"] else None), (image if cfg["image"] else None)


def compress(t, limit: int = 4096) -> np.ndarray:
    """Full tensor when small, else an evenly strided sample plus [mean, mean|.|] appended."""
    a = np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float64).reshape(-1)
    if a.size <= limit:
        return a.astype(np.float32)
    step = -(-a.size // limit)
    return np.concatenate([a[::step], [a.mean(), np.abs(a).mean()]]).astype(np.float32)


# Production-size cases (BASELINE configs[0] / configs[1] / configs[3] shapes): the packed-stream attention, the XCD
# block remap, the bound-fallback split launch and the split-K thresholds only switch on at these sizes.  Outputs are
# stored in full (fp32 roundings of the reference's fp64 run + its own fp32 run); no per-block taps.
STUDENT_BIG_CASES = {
    # exactly bench.py's image 0: production net, weights seed 0, inputs seed 1234, (10,7,7) masks, z_d ~ N(0,1)
    "prod256": dict(base_channels=64, blocks_per_level=2, cond_dim=512, num_masks=10, num_deg=10, batch=1, H=256, W=256,
                    mask_hw=(7, 7), wseed=0, xseed=1234, taps=["head"]),
}

# 512 x 512 (BASELINE configs[4] shapes; n = 65536 tokens at level 0).  The REFERENCE cannot run this size in the build
# container: its CFC attention materialises the (B, 4, n, n) score matrix (cfc.py:31-35), 68.7 GB in fp32 against 64 GB of
# RAM.  These vectors are therefore ORACLE-generated (oracle/student_ref.py in fp64, which chunks the queries of both
# attentions; the same oracle is pinned to the reference's own outputs at 256 x 256 and below): make_golden.py student_oracle
STUDENT_ORACLE_CASES = {
    "prod512": dict(base_channels=64, blocks_per_level=2, cond_dim=512, num_masks=10, num_deg=10, batch=1, H=512, W=512,
                    mask_hw=(7, 7), wseed=0, xseed=99),
}

UNET_BIG_CASES = {
    # U/options/test.yml topology and width, one 256x256 sample, float time like t * sample_scale
    "prod256": dict(nf=64, ch_mult=[1, 2, 4, 8], degra=True, image=True, batch=1, H=256, W=256, times=[37.0], wseed=8,
                    xseed=35),
}
