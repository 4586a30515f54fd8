"""``DenoisingModel`` -- the inference half of the reference's IR-SDE model wrapper.

Mirrors /root/reference/legacy/third_party/universal-image-restoration/config/daclip-sde/models/denoising_model.py:
``feed_data`` (:121-127), ``test`` (:152-162: set_mu, no_grad, reverse_sde / reverse_posterior) and
``get_current_visuals`` (:164-173) -- the three methods U/test.py:120-127, U/app.py and predict.py drive.  Training
(optimizers, EMA, schedulers, DataParallel, checkpoint I/O) is out of scope.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import torch


class DenoisingModel:
    def __init__(self, model, device="cuda") -> None:
        self.model = model      # a ConditionalUNet (the reference wraps it in DataParallel, denoising_model.py:42)
        self.device = device
        self.state = self.condition = self.state_0 = self.output = None
        self.text_context = self.image_context = None

    def feed_data(self, state, LQ, GT=None, text_context=None, image_context=None):
        self.state = state.to(self.device)
        self.condition = LQ.to(self.device)
        if GT is not None:
            self.state_0 = GT.to(self.device)
        self.text_context, self.image_context = text_context, image_context

    @torch.no_grad()
    def test(self, sde=None, mode="posterior", save_states=False, noise_tape: Optional[torch.Tensor] = None):
        sde.set_mu(self.condition)
        kw = dict(text_context=self.text_context, image_context=self.image_context)
        if mode == "sde":
            self.output = sde.reverse_sde(self.state, save_states=save_states, noise_tape=noise_tape, **kw)
        else:  # every other mode string means "posterior" in the reference (denoising_model.py:157-160)
            self.output = sde.reverse_posterior(self.state, save_states=save_states, noise_tape=noise_tape, **kw)

    def get_current_visuals(self, need_GT=True):
        out = OrderedDict()
        out["Input"] = self.condition.detach()[0].float().cpu()
        out["Output"] = self.output.detach()[0].float().cpu()
        if need_GT and self.state_0 is not None:
            out["GT"] = self.state_0.detach()[0].float().cpu()
        return out
