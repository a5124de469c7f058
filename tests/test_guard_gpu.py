"""Out-of-bounds guard: the hand-written kernels run with every input placed at the very end of its own device mapping
(tools/guard_probe.py, a child process: a GPU memory access fault aborts the process that caused it)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    ("stem", "2", "30", "600"),        # 570 tiles over 143 workgroups: the last one has two waves without a tile (the bug this test found)
    ("stem", "1", "37", "53"),
    ("stem", "1", "7", "7"),
    ("stem", "2", "64", "96"),
    ("w3", "2", "9", "97", "64"),
    ("w3", "1", "3", "30", "64"),
    ("w3", "2", "25", "34", "128"),
    ("w3", "1", "13", "21", "512"),
    ("n3", "2", "9", "97"),            # csrc/narrow3x3.hip: one strip with a masked tail
    ("n3", "1", "3", "130"),           # two strips, two pixels in the second
    ("n3", "3", "37", "257"),
    ("n3", "1", "1", "1"),
    ("dense",),
    ("pw",),
    ("pool",),
    ("bottleneck",),
    ("step", "resnet50", "200", "280"),
    ("step", "resnet18", "131", "173"),
    ("step", "resnet50", "333", "517"),
    ("step", "resnet18", "97", "400"),
    # the dense-head kernels themselves (round 4: a bf16 predict at 2 x 1344 x 1344 faulted inside rn_detect_levels)
    ("detect", "bf16", "2", "1344", "1344", "90"),
    ("detect", "bf16", "16", "1344", "1344", "90"),
    ("detect", "f16", "1", "1344", "1344", "90", "-7.0"),
    ("detect", "bf16", "3", "352", "416", "7", "-3.0"),
    ("detect", "f32", "2", "224", "160", "5", "-2.0"),
    ("detect", "bf16", "1", "352", "416", "90", "0", "1000", "500"),     # exploding head outputs (half of the candidates dead): the seg_count race
    ("detect", "bf16", "4", "1344", "1344", "90", "0", "1000", "500"),
    ("loss", "bf16", "3", "800", "1344", "90", "8"),
    ("loss", "f16", "2", "800", "1344", "90", "500"),
    ("loss", "f32", "3", "224", "160", "5", "3"),
    ("loss", "bf16", "4", "352", "416", "7", "40"),
]


@pytest.mark.parametrize("case", CASES, ids=["-".join(c) for c in CASES])
def test_kernels_stay_inside_their_operands(case):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), env.get("PYTHONPATH", "")])      # (tests/synth.py: synthetic GT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "guard_probe.py"), *case], capture_output=True, text=True, env=env,
                       timeout=600, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, f"probe {case} died (GPU memory access fault?):\n{tail}"
    assert "ok " + case[0] in r.stdout, tail
