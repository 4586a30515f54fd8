"""GPU parity of halo-tiled restoration: HIP student on batched windows + stitch (product) vs the tiling oracle
(reference-equivalent forward on each window, stitched by ownership)."""
import pytest
import torch

from oracle import student_ref as R
from oracle import tiling_ref as TR
from tests.helpers import max_abs
from underwater_image_restoration_amd.synth import student_param_shapes, synth_state_dict, synth_student_inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("with_masks", [True, False])
def test_tiled_student_matches_tiling_oracle(with_masks):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from underwater_image_restoration_amd.student import NAFNetWFIGate
    from underwater_image_restoration_amd.tiling import TiledRestorer
    M = 10 if with_masks else 0
    mk = dict(base_channels=16, num_levels=3, blocks_per_level=1, cond_dim=20, num_masks=M, num_degradation_types=None)
    sd = synth_state_dict(student_param_shapes(**mk), seed=11)
    model = NAFNetWFIGate(**mk, precision="fp32")
    model.load_state_dict(sd)
    model = model.cuda().eval()
    x, z, m = synth_student_inputs(2, 48, 80, 20, M, mask_hw=(7, 7), seed=77)   # 2 frames, 2 x 4 cells of 16 px + halo 8
    sd64 = {k: v.double() for k, v in sd.items()}

    def ref_fn(xw, c, mw):
        with torch.no_grad():
            return R.student_forward(sd64, xw, c, mw, blocks_per_level=1)[0]

    ref = TR.tiled_forward(ref_fn, x.double(), z.double(), None if m is None else m.double(), 32, 8)
    restorer = TiledRestorer(model, tile=32, halo=8, max_batch=5)
    out = restorer(x.cuda(), z.cuda(), None if m is None else m.cuda())
    e = max_abs(out.cpu().numpy(), ref.numpy())
    print(f"\n[tiled student masks={with_masks}] 48x80 frame, 32-px windows, halo 8: max abs err {e:.3e}")
    assert out.shape == x.shape and e < 1e-4
    # tiling really changes the function (global norms / attention see one window): whole-frame result differs
    whole = model(x.cuda(), z.cuda(), None if m is None else m.cuda())[0]
    assert float((whole - out).abs().max()) > 1e-3


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_inplace_windows_equal_the_gather_and_stitch_path(precision):
    """Kernel-side tiling (stem / mask-resize kernels read each window from the frame, the head kernel writes only the
    owned core into the output frame) must give exactly what gathering the windows into a batch and stitching the
    restored windows gives: same kernels, same window data, no copies."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from underwater_image_restoration_amd.student import NAFNetWFIGate
    from underwater_image_restoration_amd.tiling import TiledRestorer
    mk = dict(base_channels=16, num_levels=3, blocks_per_level=1, cond_dim=20, num_masks=10, num_degradation_types=5)
    model = NAFNetWFIGate(**mk, precision=precision)
    model.load_state_dict(synth_state_dict(student_param_shapes(**mk), seed=12))
    model = model.cuda().eval()
    x, z, m = synth_student_inputs(2, 72, 104, 20, 10, mask_hw=(7, 7), seed=78)   # ragged: last cells are clipped
    restorer = TiledRestorer(model, tile=32, halo=8, max_batch=7)
    a = restorer(x.cuda(), z.cuda(), m.cuda())
    model.supports_windows = False
    try:
        b = restorer(x.cuda(), z.cuda(), m.cuda())
    finally:
        model.supports_windows = True
    assert a.shape == x.shape and bool(torch.isfinite(a).all())
    assert torch.equal(a, b)


def test_any_size_restorer_pads_and_crops():
    """Frames whose size is not a multiple of 8: reflect-pad, restore, crop.  Aligned frames are the
    wrapped call itself; a padded frame equals the aligned pipeline run on the explicitly padded input.  (Against the
    reference's native odd-size path: parity unpinned, see AnySizeRestorer.)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from underwater_image_restoration_amd.student import NAFNetWFIGate
    from underwater_image_restoration_amd.tiling import AnySizeRestorer, TiledRestorer
    mk = dict(base_channels=16, num_levels=3, blocks_per_level=1, cond_dim=20, num_masks=10, num_degradation_types=5)
    model = NAFNetWFIGate(**mk, precision="fp32")
    model.load_state_dict(synth_state_dict(student_param_shapes(**mk), seed=12))
    model = model.cuda().eval()
    anyr = AnySizeRestorer(model, tile=32, halo=8, max_batch=8)
    x, z, m = synth_student_inputs(2, 45, 77, 20, 10, mask_hw=(7, 7), seed=5)
    out = anyr(x.cuda(), z.cuda(), m.cuda())
    assert out.shape == x.shape and bool(torch.isfinite(out).all())
    assert float(out.min()) >= 0.0 and float(out.max()) <= 1.0
    xp = torch.nn.functional.pad(x, (0, 3, 0, 3), mode="reflect")
    ref = TiledRestorer(model, tile=32, halo=8, max_batch=8)(xp.cuda(), z.cuda(), m.cuda())[:, :, :45, :77]
    assert torch.equal(out, ref)
    xa, za, ma = synth_student_inputs(1, 48, 80, 20, 10, mask_hw=(7, 7), seed=6)
    assert torch.equal(anyr(xa.cuda(), za.cuda(), ma.cuda()), TiledRestorer(model, tile=32, halo=8, max_batch=8)(xa.cuda(), za.cuda(), ma.cuda()))
    xe, ze, me = synth_student_inputs(1, 20, 28, 20, 10, mask_hw=(7, 7), seed=7)     # even, within one tile: the model itself
    assert torch.equal(anyr(xe.cuda(), ze.cuda(), me.cuda()), model(xe.cuda(), ze.cuda(), me.cuda())[0])
    with pytest.raises(ValueError):
        anyr(torch.rand(1, 3, 2, 50).cuda(), z[:1].cuda(), m[:1].cuda())
