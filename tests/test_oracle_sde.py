"""CPU: pin oracle/sde_ref.py (and the host-side tables of the product's IRSDE) against golden vectors produced by
the reference's own sde_utils.py (tests/golden/make_golden.py::sde_case)."""
import os

import numpy as np
import pytest
import torch

from oracle import sde_ref as S
from tests.helpers import max_abs


def _model(x, mu, t, **kw):
    return 0.3 * (x - mu) + 0.01 * torch.sin(x * 7 + float(t))


@pytest.mark.parametrize("sched", ["cosine", "linear", "constant"])
def test_tables_bit_exact(golden_dir, sched):
    g = np.load(os.path.join(golden_dir, "sde.npz"))
    tb = S.SdeTables(50, T=100, schedule=sched, eps=0.005)
    for nm in ("thetas", "sigmas", "thetas_cumsum", "sigma_bars"):
        assert np.array_equal(getattr(tb, nm).numpy(), g[f"{sched}_{nm}"]), nm
    assert np.float32(float(tb.dt)) == g[f"{sched}_dt"]
    # the product's IRSDE builds the same tables on the host (no GPU needed for the constructor)
    from underwater_image_restoration_amd.sde import IRSDE
    sde = IRSDE(max_sigma=50, T=100, schedule=sched, eps=0.005, device="cpu")
    for nm in ("thetas", "sigmas", "thetas_cumsum", "sigma_bars"):
        assert np.array_equal(getattr(sde, nm).numpy(), g[f"{sched}_{nm}"]), nm
    assert np.float32(float(sde.dt)) == g[f"{sched}_dt"]


def test_measured_constants():
    tb = S.SdeTables(50, T=100, schedule="cosine", eps=0.005)
    assert abs(float(tb.dt) - 0.104094) < 1e-6 and abs(float(tb.sigma_bars[100]) - 0.196076) < 1e-6


@pytest.mark.parametrize("mode", ["sde", "posterior", "ode"])
def test_trajectories_match_reference(golden_dir, mode):
    g = np.load(os.path.join(golden_dir, "sde.npz"))
    tb = S.SdeTables(50, T=100, schedule="cosine", eps=0.005)
    mu, xt = torch.from_numpy(g["traj_mu"]), torch.from_numpy(g["traj_xt"])
    tape = torch.from_numpy(g[f"traj_{mode}_tape"]) if f"traj_{mode}_tape" in g.files else None
    out = tb.sample(_model, xt, mu, mode, tape, T=6)
    assert max_abs(out.numpy(), g[f"traj_{mode}_out"]) < 2e-6


def test_unknown_schedule_raises():
    with pytest.raises(ValueError):
        S.theta_schedule("nope", 10)
