"""CPU: pin the oracle (oracle/student_ref.py) against golden vectors produced by the reference itself."""
import os

import numpy as np
import pytest
import torch

from oracle import student_ref as R
from tests.helpers import STUDENT_CASES, compress, max_abs, oracle_tap_lookup, student_case_setup

TOL = 1e-4  # fp32 oracle vs fp32 reference: same math, different summation order (net amplifies rounding)


def test_dwt_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "dwt.npz"))
    for tag in ("even", "odd"):
        x = torch.from_numpy(g[f"{tag}_x"])
        bands = R.haar_dwt2(x)
        for nm, b in zip(("ll", "lh", "hl", "hh"), bands):
            assert max_abs(b.numpy(), g[f"{tag}_{nm}"]) < 1e-6, (tag, nm)
        assert max_abs(R.haar_idwt2(*bands).numpy(), g[f"{tag}_rec"]) < 1e-6
    with pytest.raises(ValueError):
        R.haar_dwt2(torch.zeros(3, 4, 4))


@pytest.mark.parametrize("name", list(STUDENT_CASES))
def test_student_oracle_matches_reference(name):
    cfg, mk, sd, x, z, m, gold = student_case_setup(name)
    taps = {}
    with torch.no_grad():
        out, alphas, logits = R.student_forward(sd, x, z, m, blocks_per_level=cfg["blocks_per_level"], taps=taps)
    assert max_abs(out.numpy(), gold["out"]) < TOL
    assert max_abs(taps["pre_clamp"].numpy(), gold["pre_clamp"]) < TOL
    assert len(alphas) == 5 * cfg["blocks_per_level"]
    for i, a in enumerate(alphas):
        assert max_abs(compress(a), gold[f"alpha{i}"]) < 5e-4, i  # deep intermediates carry more fp32 noise
    if cfg["num_deg"] is None:
        assert logits is None
    else:
        assert max_abs(logits.numpy(), gold["logits"]) < 1e-4
    for key, val in gold.items():
        if key.startswith("tap:") and key != "tap:head":
            got = compress(oracle_tap_lookup(taps, key[4:]))
            scale = max(1.0, float(np.max(np.abs(val))))
            assert max_abs(got, val) < 5e-4 * scale, key


@pytest.mark.parametrize("name", list(STUDENT_CASES))
def test_student_oracle_math_matches_reference_fp64(name):
    """Same modules run in float64 on both sides: agreement to 1e-9 shows the oracle's math IS the reference's."""
    cfg, mk, sd, x, z, m, gold = student_case_setup(name)
    sd64 = {k: v.double() for k, v in sd.items()}
    taps = {}
    with torch.no_grad():
        out, alphas, _ = R.student_forward(sd64, x.double(), z.double(), None if m is None else m.double(),
                                           blocks_per_level=cfg["blocks_per_level"], taps=taps)
    assert max_abs(out.numpy(), gold["out_f64"]) < 1e-9
    assert max_abs(taps["pre_clamp"].numpy(), gold["pre_clamp_f64"]) < 1e-9
    assert max_abs(compress(alphas[-1]).astype(np.float64), gold["alpha_last_f64"]) < 1e-6


def test_attention_chunking_is_a_memory_device_only():
    torch.manual_seed(3)
    q, k, v = (torch.randn(1, 2, 8, 50) for _ in range(3))
    (a,) = R._global_attention(q, k, [v], chunk=7)
    (b,) = R._global_attention(q, k, [v], chunk=1024)
    assert max_abs(a.numpy(), b.numpy()) < 1e-6


def test_cond_required():
    with pytest.raises(ValueError):
        R.student_forward({}, torch.zeros(1, 3, 8, 8), None)


def test_student_oracle_matches_reference_at_production_size():
    """One 256x256 image through the full-width net (bench.py's CPU-baseline shape): the fp32 oracle against the
    reference's own fp64 and fp32 outputs.  ~1 minute of CPU: the only production-size pin of the oracle."""
    cfg, mk, sd, x, z, m, gold = student_case_setup("prod256")
    taps = {}
    with torch.no_grad():
        out, alphas, logits = R.student_forward(sd, x, z, m, blocks_per_level=cfg["blocks_per_level"], taps=taps)
    assert max_abs(out.numpy(), gold["out_f64"]) < TOL
    assert max_abs(out.numpy(), gold["out"]) < 2 * TOL
    assert max_abs(taps["pre_clamp"].numpy(), gold["pre_clamp_f64"]) < TOL
    assert max_abs(logits.numpy(), gold["logits_f64"]) < 1e-4


def test_prod512_golden_file_is_oracle_generated_and_sane():
    """tests/golden/student_prod512.npz (512 x 512, configs[4] shapes) is ORACLE-generated in fp64 (the reference's CFC
    cannot run that size in this container: tests/golden/cases.py); re-running the generator takes ~20 minutes, so here
    only the file's contract is checked: loads without pickle, the restored image is the clamp of the stored pre-clamp
    map, and the recorded fp32-vs-fp64 noise of the oracle itself is the figure the GPU test's tolerance argues from."""
    import os
    import numpy as np
    from tests.helpers import GOLDEN
    g = np.load(os.path.join(GOLDEN, "student_prod512.npz"), allow_pickle=False)
    assert str(g["generated_by"]) == "oracle/student_ref.py fp64"
    pre = g["pre_clamp_f64"]
    assert pre.shape == (1, 3, 512, 512) and pre.dtype == np.float32 and np.isfinite(pre).all()
    out = np.clip(pre, 0.0, 1.0)
    assert 0.01 < float(((out == 0) | (out == 1)).mean()) < 0.5       # not clamp-saturated: the comparison means something
    mx, p9999, mean = (float(v) for v in g["oracle_fp32_vs_fp64_max_p9999_mean"])
    assert 5e-5 < mx < 2e-4 and p9999 < 1e-4 and mean < 1e-5
    assert g["logits_f64"].shape == (1, 10)
