"""A seeded slice of every randomised cross-check (scripts/fuzz_*.py) inside the suite: fixed seeds, fixed case counts, one
process.  The scripts' longer runs found the one real bug of round 3 (a cart-pole closed loop with N = 1 never stepped its
plant) that 143 hand-written tests had missed; a slice of each now runs with every `pytest -m gpu`.

Each script compares two implementations that must agree (bit for bit where they share code, to a stated round-off bound
where they do not) on randomly drawn shapes and exits non-zero on any mismatch:
  fuzz_device_loops  persistent solve / MPC kernels vs the host-driven loops (batch 1-1024, horizon 1-90, caps, warm / cold)
  fuzz_kernels       fused sweep vs records (+ t_start, active masks), one-call iteration vs separate calls, user-model loops
  fuzz_rollouts      fused line search vs candidate rollouts (1-8 step sizes), simulate vs total cost, pack / unpack
  fuzz_transformer   fused transformer kernel vs the layer-wise fp32 kernels, gains mode vs plain output
  fuzz_train         training step vs fp32 autograd
"""
import os
import runpy
import sys

import pytest

from conftest import ROOT

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("script,seed,cases", [("fuzz_device_loops", 41, 200), ("fuzz_kernels", 42, 200), ("fuzz_rollouts", 43, 200),
                                               ("fuzz_transformer", 44, 120), ("fuzz_train", 45, 60)])
def test_seeded_fuzz_slice(script, seed, cases, monkeypatch, capsys):
    monkeypatch.setattr(sys, "argv", [script + ".py", "600", str(seed), str(cases)])
    with pytest.raises(SystemExit) as done:
        runpy.run_path(os.path.join(ROOT, "scripts", script + ".py"), run_name="__main__")
    out = capsys.readouterr().out
    print(out.strip().splitlines()[-1] if out.strip() else "(no output)")
    assert done.value.code == 0, out[-3000:]
    assert f"done: {cases} cases" in out or "done:" in out
