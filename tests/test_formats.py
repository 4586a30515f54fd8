"""CPU: pseudo-label / checkpoint formats (train_student.py:98-138,368-370; infer_adapt.py:146-160)."""
import numpy as np
import pytest
import torch

from underwater_image_restoration_amd import formats


def test_pt_and_npy_pseudo_labels(tmp_path):
    d = {"z_d": torch.randn(20), "masks": torch.rand(10, 7, 7), "global_prob": torch.rand(10)}
    (tmp_path / "sub").mkdir()
    torch.save(d, tmp_path / "sub" / "img001.pt")
    got = formats.load_pseudo_label(tmp_path, "sub/img001.png")
    assert set(got) == set(d) and all(torch.equal(got[k], d[k]) for k in d)
    assert formats.infer_model_dims(got) == (20, 10, 10) and formats.infer_model_dims(got, 4) == (20, 10, 4)
    # V2 triple: float64 arrays on disk come back float32, defaults filled in
    np.save(tmp_path / "img002_features.npy", np.random.rand(512))
    np.save(tmp_path / "img002_masks.npy", np.random.rand(10, 224, 224).astype(np.float32))
    np.save(tmp_path / "img002_probs.npy", np.random.rand(10))
    got = formats.load_pseudo_label(tmp_path, "img002.jpg")
    assert got["z_d"].dtype == torch.float32 and got["z_d"].shape == (512,) and got["masks"].shape == (10, 224, 224)
    assert torch.all(got["confidence"] == 0.8) and float(got["confidence_scale"]) == pytest.approx(0.2)
    assert formats.infer_model_dims(got) == (512, 10, 10)
    with pytest.raises(FileNotFoundError, match="img003"):
        formats.load_pseudo_label(tmp_path, "img003.png")


def test_pt_files_are_loaded_without_executing_anything(tmp_path):
    class Evil:
        def __reduce__(self):
            return (print, ("should never run",))
    torch.save({"z_d": Evil()}, tmp_path / "bad.pt")
    with pytest.raises(Exception):
        formats.load_pseudo_label(tmp_path, "bad.png")
