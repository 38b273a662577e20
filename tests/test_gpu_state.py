"""Step-state publication (eg_set_step_state): the host may queue many steps without synchronising; every queued step must
still see ITS OWN seed / lr / bias corrections.  Regression test for the pinned-staging-buffer race the round-1 advisor
found (a later step's memmove overwrote the buffer before an earlier step's H2D copy had run)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from eyegaze_multimodal_amd import HipAdamW  # noqa: E402
from eyegaze_multimodal_amd.data import randn_windows  # noqa: E402
from tests.test_gpu_model import DEV, build  # noqa: E402


def _run(sync_every_step: bool, steps: int = 10, stall: bool = False):
    z, kw, cfg, sd, model = build("cfg3_xattn", "bf16")
    model.train()
    B = 16
    x1, x2, y = (a.to(DEV) for a in randn_windows(B, 8, 1024, seed=5, num_classes=3))
    eng = model.engine(B, 1024, DEV)
    opt = HipAdamW(model, lr=3e-4)
    one = torch.ones(1, device=DEV)
    big = torch.randn(4096, 4096, device=DEV)
    norms = []
    if stall:      # keep the device busy so that the host really runs ahead of it
        for _ in range(20):
            big = big @ big * 1e-3
    for i in range(steps):
        opt.lr = 3e-4 * (1.0 + 0.1 * i)            # a different lr every step: a stale state word changes the result
        opt.begin_step(eng, seed=100 + i)
        eng.forward(x1, x2, y, train=True)
        eng.backward(gloss=one)
        opt.step(eng)
        norms.append(eng.state_dev.clone())         # device-side snapshot, stream ordered, no host sync
        if sync_every_step:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return model._flat.flat.clone(), torch.stack(norms).cpu()


def test_unsynchronised_steps_equal_synchronised_steps():
    p_sync, s_sync = _run(True)
    p_free, s_free = _run(False, stall=True)
    assert torch.equal(s_sync, s_free), "a queued step read another step's state words"
    assert torch.equal(p_sync, p_free)
    # every step carried its own seed and lr
    assert len({int(v) for v in s_free[:, 0]}) == s_free.shape[0]
    assert len({int(v) for v in s_free[:, 2]}) == s_free.shape[0]


def test_engine_has_no_host_staging_buffer():
    z, kw, cfg, sd, model = build("cfg2_concat", "bf16")
    eng = model.engine(4, 1024, DEV)
    assert not hasattr(eng, "state_host")
    st = eng.read_state()
    assert st.scaler_on == 0 and st.loss_scale == 1.0 and st.found_inf == 0
