"""Opt-in soak (NMFK_TEST_BURNER=1): every shipped path repeated while ANOTHER process keeps gfx950's 128-bit-operand
matrix instructions busy on every CU -- the neighbour that exposed the packed-fp32 hazard of DESIGN.md ("Known hazard").
Each repetition must reproduce, bit for bit, a reference computed with the GPU to ourselves.  Skipped by default: it
starts a second GPU process (tools/hazard/burner, built with hipcc on the spot) and runs for about a minute; the generated-code
lint of tests/test_isa_lint.py is the always-on guard.  profiles/r02/soak_beside_burner.txt has the full-length run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.skipif(not os.environ.get("NMFK_TEST_BURNER"), reason="opt-in: NMFK_TEST_BURNER=1 (starts a second GPU process)")
def test_shipped_paths_reproduce_beside_a_wide_operand_mfma_neighbour():
    env = dict(os.environ, REPS_SCALE=os.environ.get("REPS_SCALE", "0.5"), LIMIT="400")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "hazard", "soak_beside_burner.sh"), "0", "300"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert "TOTAL differing: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
